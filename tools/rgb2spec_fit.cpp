// rgb2spec_fit — regenerates the sRGB -> sigmoid-polynomial coefficient table in the reference's
// binary layout  [64 f32 z_nodes][3][64][64][64][3] f32  (rgb_to_spec/src/lib.rs:1-4,
// spectrum/src/rgb_sigmoid_polynomial.rs:35-84).  The reference's own table is a git-LFS object that
// is absent here, and its fitter (rgb_to_spec/python/main.py) is a stochastic Adam optimisation; this
// is a deterministic damped Newton solve of the same model:
//     S(lambda) = sigmoid(c0 t^2 + c1 t + c2),  t = (lambda-360)/470          (rgb_sigmoid_polynomial.rs:179-182)
//     rgb(S)    = M_xyz->srgb * sum_lambda S * D65n * (xbar,ybar,zbar)          (main.py:192-199)
// on the grid  z = smoothstep(smoothstep(k/63)), x = xi/63*z, y = yi/63*z with max-component
// ordering table[m][zi][yi][xi]                                                (main.py:53-58,165-175)
// Coefficients are box-limited like the reference's tanh-scaled parameters (main.py:67-74,186-190).
//
// usage: rgb2spec_fit <presets470.bin> <idx_x> <idx_y> <idx_z> <idx_d65> <out.bin> [threads]
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static const int N = 470, RES = 64;
static double W[3][N];   // rgb weights per wavelength: M * (cmf * d65n)
static double T1[N], T2[N];

static inline double smoothstep(double x) { return x * x * (3.0 - 2.0 * x); }

static void eval(const double c[3], double rgb[3], double J[3][3]) {
    for (int i = 0; i < 3; ++i) { rgb[i] = 0; for (int j = 0; j < 3; ++j) J[i][j] = 0; }
    for (int k = 0; k < N; ++k) {
        double x = c[0] * T2[k] + c[1] * T1[k] + c[2];
        double s = 1.0 / (1.0 + std::exp(-x));
        double ds = s * (1.0 - s);
        for (int i = 0; i < 3; ++i) {
            double w = W[i][k];
            rgb[i] += w * s;
            J[i][0] += w * ds * T2[k]; J[i][1] += w * ds * T1[k]; J[i][2] += w * ds;
        }
    }
}
static bool solve3(double A[3][3], const double b[3], double x[3]) {
    double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                 A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    if (!(std::fabs(det) > 1e-300)) return false;
    double inv = 1.0 / det;
    x[0] = inv * (b[0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (b[1] * A[2][2] - A[1][2] * b[2]) + A[0][2] * (b[1] * A[2][1] - A[1][1] * b[2]));
    x[1] = inv * (A[0][0] * (b[1] * A[2][2] - A[1][2] * b[2]) - b[0] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) + A[0][2] * (A[1][0] * b[2] - b[1] * A[2][0]));
    x[2] = inv * (A[0][0] * (A[1][1] * b[2] - b[1] * A[2][1]) - A[0][1] * (A[1][0] * b[2] - b[1] * A[2][0]) + b[0] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]));
    return true;
}
static const double LIM01 = 400.0, LIM2 = 60.0;
static void clampc(double c[3]) {
    for (int i = 0; i < 2; ++i) c[i] = std::fmax(-LIM01, std::fmin(LIM01, c[i]));
    c[2] = std::fmax(-LIM2, std::fmin(LIM2, c[2]));
}
static double fit(const double target[3], double c[3]) {
    double rgb[3], J[3][3];
    eval(c, rgb, J);
    auto err2 = [&](const double r[3]) { double e = 0; for (int i = 0; i < 3; ++i) e += (r[i] - target[i]) * (r[i] - target[i]); return e; };
    double e = err2(rgb);
    double mu = 1e-9;
    for (int it = 0; it < 60 && e > 1e-16; ++it) {
        // Levenberg-Marquardt on the 3x3 system
        double A[3][3], g[3];
        for (int i = 0; i < 3; ++i) {
            g[i] = 0;
            for (int j = 0; j < 3; ++j) { A[i][j] = 0; for (int k = 0; k < 3; ++k) A[i][j] += J[k][i] * J[k][j]; }
            for (int k = 0; k < 3; ++k) g[i] -= J[k][i] * (rgb[k] - target[k]);
        }
        bool improved = false;
        for (int tries = 0; tries < 12; ++tries) {
            double B[3][3];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[i][j] = A[i][j] + (i == j ? mu * (A[i][i] + 1e-12) : 0.0);
            double d[3];
            if (solve3(B, g, d)) {
                double cn[3] = {c[0] + d[0], c[1] + d[1], c[2] + d[2]};
                clampc(cn);
                double rn[3], Jn[3][3];
                eval(cn, rn, Jn);
                double en = err2(rn);
                if (en < e) {
                    for (int i = 0; i < 3; ++i) { c[i] = cn[i]; rgb[i] = rn[i]; for (int j = 0; j < 3; ++j) J[i][j] = Jn[i][j]; }
                    e = en; mu = std::fmax(mu * 0.2, 1e-12); improved = true;
                    break;
                }
            }
            mu *= 8.0;
        }
        if (!improved) break;
    }
    return e;
}

int main(int argc, char** argv) {
    if (argc < 7) { std::fprintf(stderr, "usage: %s presets.bin ix iy iz id65 out.bin [threads]\n", argv[0]); return 2; }
    int ix = std::atoi(argv[2]), iy = std::atoi(argv[3]), iz = std::atoi(argv[4]), id = std::atoi(argv[5]);
    int nthreads = argc > 7 ? std::atoi(argv[7]) : 8;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("presets"); return 1; }
    std::vector<float> all;
    { float buf[N]; while (std::fread(buf, sizeof(float), N, f) == (size_t)N) all.insert(all.end(), buf, buf + N); }
    std::fclose(f);
    auto lut = [&](int i) { return all.data() + (size_t)i * N; };
    // sRGB matrix from primaries (color/src/gamut.rs:29-63), double precision
    auto xyz = [](double x, double y, double o[3]) { o[0] = x / y; o[1] = 1.0; o[2] = (1.0 - x - y) / y; };
    double r[3], g[3], b[3], w[3];
    xyz(0.64, 0.33, r); xyz(0.30, 0.60, g); xyz(0.15, 0.06, b); xyz(0.3127, 0.3290, w);
    double P[3][3] = {{r[0], g[0], b[0]}, {r[1], g[1], b[1]}, {r[2], g[2], b[2]}};
    double cw[3]; solve3(P, w, cw);
    double R2X[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R2X[i][j] = P[i][j] * cw[j];
    double M[3][3];   // inverse of R2X via solving for unit vectors
    for (int j = 0; j < 3; ++j) { double e[3] = {0, 0, 0}; e[j] = 1; double col[3]; solve3(R2X, e, col); for (int i = 0; i < 3; ++i) M[i][j] = col[i]; }
    for (int k = 0; k < N; ++k) {
        double t = (double)k / 470.0;   // lambda = 360 + k
        T1[k] = t; T2[k] = t * t;
        double d = lut(id)[k];
        double cm[3] = {lut(ix)[k] * d, lut(iy)[k] * d, lut(iz)[k] * d};
        for (int i = 0; i < 3; ++i) W[i][k] = M[i][0] * cm[0] + M[i][1] * cm[1] + M[i][2] * cm[2];
    }
    std::vector<float> out(RES + 3 * (size_t)RES * RES * RES * 3);
    double zn[RES];
    for (int k = 0; k < RES; ++k) { zn[k] = smoothstep(smoothstep((double)k / (RES - 1))); out[k] = (float)zn[k]; }
    std::atomic<int> next{0};
    std::atomic<int> bad{0};
    double worst = 0;
    auto worker = [&]() {
        for (;;) {
            int job = next.fetch_add(1);
            if (job >= 3 * RES * RES) break;
            int m = job / (RES * RES), yi = (job / RES) % RES, xi = job % RES;
            double x = (double)xi / (RES - 1), y = (double)yi / (RES - 1);
            auto cell = [&](int zi) { return out.data() + RES + ((((size_t)m * RES + zi) * RES + yi) * RES + xi) * 3; };
            const int start = RES / 5;
            for (int dir = 0; dir < 2; ++dir) {
                double c[3] = {0, 0, 0};
                for (int zi = start; dir == 0 ? zi < RES : zi >= 0; zi += dir == 0 ? 1 : -1) {
                    if (dir == 1 && zi == start) { continue; }
                    double z = zn[zi];
                    double tgt[3]; tgt[m] = z; tgt[(m + 1) % 3] = x * z; tgt[(m + 2) % 3] = y * z;
                    if (dir == 1 && zi == start - 1) { float* p = cell(start); c[0] = p[0]; c[1] = p[1]; c[2] = p[2]; }
                    double e = fit(tgt, c);
                    if (e > 1e-6) bad++;
                    float* p = cell(zi);
                    p[0] = (float)c[0]; p[1] = (float)c[1]; p[2] = (float)c[2];
                }
            }
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nthreads; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    (void)worst;
    std::fprintf(stderr, "rgb2spec_fit: %d cells with squared rgb residual > 1e-6 (gamut-boundary cells)\n", bad.load());
    FILE* o = std::fopen(argv[6], "wb");
    if (!o) { std::perror("out"); return 1; }
    std::fwrite(out.data(), sizeof(float), out.size(), o);
    std::fclose(o);
    return 0;
}
