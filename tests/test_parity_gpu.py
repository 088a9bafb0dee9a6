"""GPU parity tests: the HIP product (through the C ABI) against the oracle on the same seeded inputs.

Bars: integer work (Sobol) bit-exact; floating point within the tolerances written in each test
(BASELINE north_star: "output matches ... within a stated per-pixel L2 tolerance, Sobol bit-exact vs CPU").
"""
import os

import numpy as np
import pytest

from conftest import gamma22_rmse_u8, linear_rmse_u8, median3

pytestmark = pytest.mark.gpu


# --------------------------------------------------------------------------------------- Sobol: bit-exact
@pytest.mark.parametrize("w,h,spp", [(256, 256, 16), (1920, 1080, 1024), (1920, 1080, 4096), (4096, 4096, 1024),
                                      (1920, 1080, 16384), (200, 150, 512), (200, 150, 2048), (64, 48, 1)])
def test_sobol_bit_exact(product, oracle, w, h, spp):
    rng = np.random.default_rng(spp + w)
    n = 20000
    xys = np.stack([rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, spp, n)], 1).astype(np.uint32)
    xys[:4] = [[0, 0, 0], [w - 1, h - 1, spp - 1], [w - 1, 0, 0], [0, h - 1, spp - 1]]
    # the integrator's draw pattern: 1,2 then per bounce 1,2,(1,1,2),(1)
    pattern = "12" + "121121" * 6 + "1212"
    for seed in (0, 12345):
        a = product.probe_sobol(w, h, spp, seed, xys, pattern)
        b = oracle.probe_sobol(w, h, spp, seed, xys, pattern)
        assert np.array_equal(a, b)


# --------------------------------------------------------------------------------------- scenes
@pytest.fixture(scope="module")
def scenes3(product, oracle, pkg):
    out = {}
    for name, be in (("gpu", product), ("cpu", oracle)):
        sc = be.new_scene()
        cam = pkg.scenes.load_scene(sc, 3, 256, 256, tex_size=256)
        out[name] = (sc, cam)
    oracle.set_faithful(out["cpu"][0], False)
    return out


def _camera_rays(n, seed, spread=0.45):
    rng = np.random.default_rng(seed)
    d = np.stack([rng.uniform(-spread, spread, n), rng.uniform(-0.55, 0.1, n), -np.ones(n)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.zeros((n, 3), np.float32), d.astype(np.float32)


def test_closest_hit_parity(scenes3):
    """Scene::intersect: same triangle for >= 99.99 % of rays and |dt| <= 4 ulp-ish (2e-6 relative) on those."""
    o, d = _camera_rays(200000, 1)
    tg, ig, trg, ng = scenes3["gpu"][0].probe_intersect(o, d)
    tc, ic, trc, nc = scenes3["cpu"][0].probe_intersect(o, d)
    assert (tg > 0).mean() > 0.99
    same = (ig == ic) & (trg == trc)
    assert same.mean() >= 0.9999, same.mean()
    rel = np.abs(tg[same] - tc[same]) / np.maximum(np.abs(tc[same]), 1e-6)
    assert rel.max() <= 2.5e-7, rel.max()       # (round 3: the probe reports the reference's local-space t of the triangle found: the same float, up to the rare render-space fallback)
    assert np.abs(ng[same] - nc[same]).max() <= 2.5e-7


def test_secondary_ray_hit_parity(scenes3):
    """Incoherent rays from surface points (what bounces look like)."""
    o, d = _camera_rays(50000, 2)
    t, inst, tri, n = scenes3["cpu"][0].probe_intersect(o, d)
    ok = t > 0
    p = o[ok] + d[ok] * t[ok, None]
    rng = np.random.default_rng(5)
    d2 = rng.normal(size=p.shape).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    nn = n[ok]
    flip = np.sum(d2 * nn, 1) * np.sum(-d[ok] * nn, 1) < 0      # keep them on the side the ray came from
    d2[flip] *= -1
    o2 = (p + d2 * 1e-3).astype(np.float32)
    tg, ig, trg, _ = scenes3["gpu"][0].probe_intersect(o2, d2)
    tc, ic, trc, _ = scenes3["cpu"][0].probe_intersect(o2, d2)
    same = (ig == ic) & (trg == trc)
    assert same.mean() >= 0.9995, same.mean()
    both = same & (tc > 0)
    rel = np.abs(tg[both] - tc[both]) / np.maximum(np.abs(tc[both]), 1e-3)
    # grazing rays are ill-conditioned in t (t = distance / cos): bound the bulk tightly, the tail loosely
    assert np.percentile(rel, 99.9) <= 5e-5, np.percentile(rel, 99.9)
    assert np.median(rel) <= 2e-7
    # any-hit agrees with closest-hit on BOTH sides: within t_max there is an occluder exactly when that side's own closest hit lies
    # within t_max.  (Round 1 compared the GPU's any-hit with the ORACLE's closest-hit distance and allowed three mismatches: the one
    # that occurs — tools/anyhit_probe.py, ray 48787 — is a ray 1e-6 above the wall it starts on, grazing at cos 1e-3; both sides hit
    # the same triangle, the GPU at t = 5.6e-4, the oracle at 1.8e-3, because the reference moves the ray into the primitive's local
    # space first (origin + camera position, one rounding of 5e-7 on a distance of 1e-6, primitive/impls/triangle_mesh.rs:97) while the
    # product intersects render-space triangles.  Each side is consistent with itself, which is what any-hit has to guarantee.)
    tm = np.where(tc > 0, tc * 0.5, 1e30).astype(np.float32)
    og, oc = scenes3["gpu"][0].probe_occluded(o2, d2, tm), scenes3["cpu"][0].probe_occluded(o2, d2, tm)
    # (Round 3: the closest-hit probe now reports the reference's LOCAL-space t of the triangle found — winner_hit, pt_device.hpp — while the
    # any-hit traversal, a yes / no answer, stays in render space: on that one grazing ray the GPU's two answers now differ like GPU and
    # oracle used to.  At most two such rays in 50 000.)
    assert np.count_nonzero((og != 0) != ((tg > 0) & (tg <= tm))) <= 2
    assert np.array_equal(oc != 0, (tc > 0) & (tc <= tm))
    well = both & (np.abs(tg - tc) <= 1e-3 * np.abs(tc))                       # rays whose hit distance is well conditioned
    assert np.count_nonzero(og[well] != oc[well]) <= 2      # (the same grazing ray: its closest-hit distances now agree, its any-hit answers do not)
    tm2 = np.where(tc > 0, tc * 1.5, 1e30).astype(np.float32)
    og = scenes3["gpu"][0].probe_occluded(o2, d2, tm2)
    oc = scenes3["cpu"][0].probe_occluded(o2, d2, tm2)
    assert np.count_nonzero((og != 0) != ((tg > 0) & (tg <= tm2))) <= 2 and np.array_equal(oc != 0, (tc > 0) & (tc <= tm2))
    assert (og == oc).mean() >= 0.9999 and og.sum() >= (tc > 0).sum() * 0.999


@pytest.mark.parametrize("strategy", ["pt", "nee", "mis"])
def test_per_sample_radiance_parity(scenes3, product, pkg, strategy):
    """BaseSrgbRenderer::render's per-sample spectral radiance, same (pixel, sample) queries, Sobol sampler.
    Tolerance: >= 99 % of the samples agree to 1e-3 relative (+1e-4 absolute); the rest are paths that flipped at
    a geometric discontinuity because GPU and CPU round differently (FMA contraction, libm)."""
    rng = np.random.default_rng(11)
    n = 40000
    xys = np.stack([rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 64, n)], 1).astype(np.uint32)
    prm = pkg.make_params(64, strategy, "sobol")
    Lg, lg, pg = scenes3["gpu"][0].probe_radiance(scenes3["gpu"][1], prm, xys)
    Lc, lc, pc = scenes3["cpu"][0].probe_radiance(scenes3["cpu"][1], prm, xys)
    assert np.array_equal(lg, lc)              # wavelengths come straight from Sobol bits
    assert np.array_equal(pg, pc)
    close = np.all(np.abs(Lg - Lc) <= 1e-3 * np.abs(Lc) + 1e-4, axis=1)
    assert close.mean() >= 0.9995, close.mean()
    assert abs(Lg.mean() - Lc.mean()) <= 1e-3 * Lc.mean()


def test_image_parity_scene3_mis_sobol(scenes3, product, oracle, pkg):
    """Config 1 resolution (256x256x16, scene 3) with the headline integrator: image RMSE (the reference's own
    metric, regression_test.rs:6-40) <= 0.004 — an order of magnitude under the reference's 0.05 pass bar."""
    prm = pkg.make_params(16, "mis", "sobol")
    img_g = product.render(scenes3["gpu"][0], scenes3["gpu"][1], prm)
    img_c = oracle.render(scenes3["cpu"][0], scenes3["cpu"][1], prm)
    qg, qc = product.quantize_u8(img_g), oracle.quantize_u8(img_c)
    assert linear_rmse_u8(qg, qc) <= 0.001
    assert (np.abs(qg.astype(int) - qc.astype(int)) <= 1).mean() >= 0.999


# Frame bar of the sample-for-sample test: (rmse of the tone-mapped frames, pixels off by more than 0.01).  Round 3 made the shading point
# the reference's bit for bit (winner_hit / load_surface / shading_frames_numeric / ggx_D / pt_libm.hpp), so there are no per-scene limits and no
# relaxed roulette gate any more: measured 4e-8 ... 8.5e-5 and 0 pixels for all 49 pairs (profiles/r03_frame_table.jsonl; round 2 needed up
# to 0.03 / 150 pixels on the rough-refraction scenes and rr_gate_slack on the solid-plastic ones).  What is left below that bar: half of the
# samples are bit-equal, the other half differ by an ulp of their radiance (a reciprocal shared by four wavelengths, the hardware exp2 / rcp of
# the albedo sigmoid: two build options remove both, 0.9999 bit-equal), and 8 - 20 of a frame's 196 608 samples differ by more
# in radiances of 1e-5 and below: light connections at a grazing angle to the light, whose any-hit answer in the reference depends on the
# box tests of ITS OWN two-level BVH (tools/bit_equal_share.py, tools/top_diff.py, profiles/r03_bit_exact_options.log).
FRAME_BAR = (1.5e-4, 0)
# share of 30 000 samples whose spectral radiance agrees with the oracle's to 1e-3 (test_other_scenes_radiance_parity): measured >= 0.99993
# for every scene / strategy pair (profiles/r03_per_sample_rates.jsonl; round 2's bar was 0.98, round 3's first 0.993 ... 0.999)
PER_SAMPLE_MIN = 0.9995


@pytest.mark.parametrize("scene_id,strategy", [(0, "mis"), (0, "nee"), (1, "nee"), (2, "mis"), (3, "mis"), (3, "nee"), (3, "pt"), (4, "mis"), (5, "pt"),
                                               (5, "nee"), (6, "mis"), (7, "mis"), (7, "nee"), (8, "mis"), (9, "mis"), (10, "mis"), (11, "nee"), (11, "mis"),
                                               (12, "mis"), (13, "mis"), (14, "nee"), (15, "mis"), (16, "mis"), (17, "nee"), (17, "mis"), (18, "nee"),
                                               (19, "mis"), (19, "pt"), (19, "nee"), (20, "mis"), (20, "pt"), (21, "mis"), (21, "nee"), (22, "mis"),
                                               (22, "nee"), (27, "mis"), (27, "nee"), (27, "pt"),
                                               (29, "pt"), (29, "nee"), (29, "mis"), (30, "pt"), (30, "nee"), (30, "mis"),
                                               (31, "pt"), (31, "nee"), (31, "mis"), (32, "nee"), (32, "mis")])
def test_frames_match_the_oracle_sample_for_sample(product, oracle, pkg, scene_id, strategy):
    """Every scene id of the radiance test, through the scene's own kernel specialisation (= what bench.py runs for it): GPU and oracle
    trace the SAME paths, with the reference's own Russian-roulette gate, in every scene — rough refraction (scenes 11, 12, 27: round 2
    compared them on widened limits and not at all under MIS), near-mirror rough metal (7) and the solid constant-eta plastic heroes (9, 13,
    19: round 2 compared them with the gate relaxed on both sides) included.  At 64 spp the tone-mapped frames agree to 1.5e-4 RMSE with no
    pixel off by more than 0.01 (FRAME_BAR above: what is left below that, and why).  What it took is in DESIGN.md 2.1: the shading point rebuilt the reference's way (local-space
    intersection, numeric matrix inverses), the reference's own ill-conditioned GGX expressions, the host libm's sin / cos."""
    pair = {}
    for name, be in (("gpu", product), ("cpu", oracle)):
        sc = be.new_scene()
        pair[name] = (sc, pkg.scenes.load_scene(sc, scene_id, 64, 48, tex_size=128))
    oracle.set_faithful(pair["cpu"][0], False)
    prm = pkg.make_params(64, strategy, "sobol")
    g = product.render(pair["gpu"][0], pair["gpu"][1], prm)
    c = oracle.render(pair["cpu"][0], pair["cpu"][1], prm)
    with np.errstate(invalid="ignore"):
        assert np.array_equal(np.isnan(g), np.isnan(c))          # (the reference accumulates NaN samples, sensor.rs:42: the same pixels on both sides)
        d = np.nan_to_num(g - c)
    rmse = float(np.sqrt(np.mean(d ** 2)))
    off = int((np.abs(d).max(axis=2) > 0.01).sum())
    if os.environ.get("MI355PT_FRAME_LOG"):
        with open(os.environ["MI355PT_FRAME_LOG"], "a") as f:
            f.write(f'{{"scene": {scene_id}, "strategy": "{strategy}", "rmse": {rmse:.3e}, "off": {off}, "nan_px": {int(np.isnan(g).any(axis=2).sum())}}}\n')
    assert rmse <= FRAME_BAR[0] and off <= FRAME_BAR[1], (rmse, off)


@pytest.mark.parametrize("scene_id,strategy", [(17, "nee"), (17, "mis"), (16, "mis"), (18, "nee"), (19, "mis")])
def test_albedo_lut_frames_match_the_oracle(product, oracle, pkg, scene_id, strategy):
    """mi355pt_params.albedo_lut (SURVEY Appendix A, Q13's option): the coat weight of SimpleClearcoatPbrMaterial from the 64-entry E(cos theta)
    table of its material instead of the 64-sample estimate.  GPU and oracle read the same table (mi355pt_coat_albedo_table) with the
    same interpolation, so they trace the same paths: the sample-for-sample frame bar.  The option changes the picture by less than the
    estimator's own noise (tools/clearcoat_modes.py, profiles/r02_clearcoat_modes.jsonl)."""
    pair = {}
    for name, be in (("gpu", product), ("cpu", oracle)):
        sc = be.new_scene()
        pair[name] = (sc, pkg.scenes.load_scene(sc, scene_id, 64, 48, tex_size=128))
    oracle.set_faithful(pair["cpu"][0], False)
    oracle.set_clearcoat_mode(pair["cpu"][0], "lut", product)
    prm = pkg.make_params(64, strategy, "sobol", albedo_lut=1)
    g = product.render(pair["gpu"][0], pair["gpu"][1], prm)
    c = oracle.render(pair["cpu"][0], pair["cpu"][1], prm)
    rmse = float(np.sqrt(np.mean((g - c) ** 2)))
    off = int((np.abs(g - c).max(axis=2) > 0.01).sum())
    assert rmse <= FRAME_BAR[0] and off <= FRAME_BAR[1], (rmse, off)
    # and the option is an option: the default (64-sample estimate) gives another, equally noisy, frame of the same scene
    d = product.render(pair["gpu"][0], pair["gpu"][1], pkg.make_params(64, strategy, "sobol"))
    assert not np.array_equal(d, g) and abs(float(d.mean()) - float(g.mean())) <= 0.01 * float(g.mean())


@pytest.mark.parametrize("scene_id", [9, 13])
def test_solid_plastic_paths_follow_the_reference_gate(product, oracle, pkg, scene_id):
    """Rounds 1-2's 'open divergence' on the solid plastic heroes (scene_9.rs, scene_13.rs), closed in round 3.  A specular REFLECTION off a
    constant-eta dielectric returns f = F and pdf = F / (F + (1 - F)) (dielectric.rs:380-466), so the throughput becomes
    T * (F * (1 / pdf)) = 1 or 1 - 1 ulp depending on the last bit of F — and apply_russian_roulette (base_renderer.rs:76-92) draws its
    random number only if max(T) < 1: an implementation whose cos(theta) differs in the last ulp takes the other branch and every later
    Sobol dimension of the sample shifts.  Round 2 proved the gate was where the paths parted (relaxing it to max(T) >= 1 - 1e-5 on both
    sides brought them together: 0.29 % / 0.09 % of the samples -> 1e-5) and compared these scenes only that way.  Round 3 removed the ulp
    instead: the product now intersects the triangle it found in the mesh's local space and carries the hit back through local_to_render
    like primitive/impls/triangle_mesh.rs:89-119, inverts the shading frame numerically like math/src/transform.rs:186-203, and computes
    sin / cos like the host's libm (pt_libm.hpp).  With the REFERENCE'S OWN gate (rr_gate_slack = 0, the only value a caller can set
    without mi355pt_debug_unlock) at most 1e-4 of the samples differ now; measured 0 of 196 608 on both scenes."""
    pair = {}
    for name, be in (("gpu", product), ("cpu", oracle)):
        sc = be.new_scene()
        pair[name] = (sc, pkg.scenes.load_scene(sc, scene_id, 64, 48, tex_size=128))
    oracle.set_faithful(pair["cpu"][0], False)
    ys, xs, ss = np.meshgrid(np.arange(48), np.arange(64), np.arange(64), indexing="ij")
    xys = np.stack([xs.ravel(), ys.ravel(), ss.ravel()], 1).astype(np.uint32)
    prm = pkg.make_params(64, "mis", "sobol")
    Lg, lg, pg = pair["gpu"][0].probe_radiance(pair["gpu"][1], prm, xys)
    Lc, lc, pc = pair["cpu"][0].probe_radiance(pair["cpu"][1], prm, xys)
    assert np.array_equal(lg, lc) and np.array_equal(pg, pc)
    share = float((~np.all(np.abs(Lg - Lc) <= 1e-3 * np.abs(Lc) + 1e-4, axis=1)).mean())
    assert share <= 1e-4, share


def test_config1_pt_random(scenes3, product, oracle, pkg):
    """BASELINE configs[0]: scene3 256x256, 16 spp, pt + random sampler.  Both sides use the same counter-based
    stream (the reference's ThreadRng is unseeded, so only statistical parity exists there)."""
    prm = pkg.make_params(16, "pt", "random")
    qg = product.quantize_u8(product.render(scenes3["gpu"][0], scenes3["gpu"][1], prm))
    qc = oracle.quantize_u8(oracle.render(scenes3["cpu"][0], scenes3["cpu"][1], prm))
    assert linear_rmse_u8(qg, qc) <= 0.01


@pytest.mark.parametrize("scene_id", [3, 5])
def test_gpu_pt_nee_mis_consistency(product, pkg, scene_id):
    """renderer_consistency_test.rs:319-353 on the GPU at the reference's own size: 200x150, 2048 spp, random — both scenes the
    reference runs (scene 3 :319-335, scene 5 :337-352)."""
    sc = product.new_scene()
    cam = pkg.scenes.load_scene(sc, scene_id, 200, 150, tex_size=256)
    imgs = {s: median3(product.quantize_u8(product.render(sc, cam, pkg.make_params(2048, s, "random")))) for s in ("pt", "nee", "mis")}
    assert gamma22_rmse_u8(imgs["pt"], imgs["nee"]) <= 0.013
    assert gamma22_rmse_u8(imgs["pt"], imgs["mis"]) <= 0.013


@pytest.mark.parametrize("scene_id", [30, 31])
def test_gpu_pt_nee_mis_consistency_textured_emitter(product, pkg, scene_id):
    """The reference's estimator-consistency criterion on the textured emitters: PT sees the panel's radiance at the HIT uv, NEE
    and MIS sample a point on it and look the radiance up THERE, weighting the light by its value at uv (0.5, 0.5) — three code paths
    that must converge to one picture.  Scene 30: Albedo-type radiance texture, constant intensity; scene 31: Illuminant-type texture
    (RgbIlluminantSpectrum per texel, rgb_texture.rs:56-64) times a FloatParameter::texture intensity ramp (emissive_material.rs:55-56)."""
    sc = product.new_scene()
    cam = pkg.scenes.load_scene(sc, scene_id, 200, 150, tex_size=256)
    # (32768 spp: unidirectional PT finds the 1.8 x 1.8 panel by chance only; 0.0146 at 8192 spp is its noise — the oracle's LINEAR channel
    # means of the three strategies agree within 1 %)
    imgs = {s: median3(product.quantize_u8(product.render(sc, cam, pkg.make_params(32768, s, "random")))) for s in ("pt", "nee", "mis")}
    assert gamma22_rmse_u8(imgs["pt"], imgs["nee"]) <= 0.013
    assert gamma22_rmse_u8(imgs["pt"], imgs["mis"]) <= 0.013


def test_gpu_pt_nee_mis_consistency_metal(product, pkg):
    """The same estimator-consistency criterion on the four rough-gold heroes of scene 7 (ConductorBsdf sample / evaluate /
    pdf must agree for PT, NEE and MIS to converge to one image)."""
    sc = product.new_scene()
    cam = pkg.scenes.load_scene(sc, 7, 200, 150)
    # 8192 spp: unidirectional PT resolves the gold heroes' caustics more slowly than scene 3's diffuse transport
    imgs = {s: median3(product.quantize_u8(product.render(sc, cam, pkg.make_params(8192, s, "random")))) for s in ("pt", "nee", "mis")}
    assert gamma22_rmse_u8(imgs["pt"], imgs["nee"]) <= 0.013
    assert gamma22_rmse_u8(imgs["pt"], imgs["mis"]) <= 0.013


@pytest.mark.parametrize("scene_id,strategy", [(8, "mis"), (10, "mis"), (0, "nee"), (17, "nee"), (17, "mis"), (11, "mis"), (11, "nee"),
                                               (6, "mis"), (7, "mis"), (7, "nee"), (20, "mis"), (20, "pt"),
                                               (1, "nee"), (2, "mis"), (21, "mis"), (21, "nee"), (19, "mis"), (19, "pt"), (19, "nee"), (22, "mis"), (22, "nee"),
                                               (4, "mis"), (5, "nee"), (9, "mis"), (12, "mis"), (13, "mis"), (14, "nee"), (15, "mis"), (16, "mis"), (18, "nee"),
                                               (27, "mis"), (27, "nee"), (27, "pt"),
                                               (29, "mis"), (29, "pt"), (30, "mis"), (30, "nee")])
def test_other_scenes_radiance_parity(product, oracle, pkg, scene_id, strategy):
    """Glass (scene 8: dispersive, wavelength termination), thin plastic (scene 10), plain Lambert (scene 0), rough clearcoat
    over rough metal (scene 17), rough SF11 glass (scene 11: microfacet reflection/transmission + light connection), smooth
    gold (scene 6), four instanced rough-gold heroes (scene 7: ConductorBsdf + complex Fresnel), SimplePbrMaterial with
    mixed metallic (scene 20, not a reference scene), point lights only (scenes 1, 2), spot + directional + area light
    together (scene 21, not a reference scene), environment light over SimplePbr / clearcoat / plastic heroes (scene 19), SimplePbr with FloatTexture metallic / roughness maps,
    textured base colour and a normal map (scenes 15 and 22), and the remaining Cornell scenes of the reference: 4 / 5 (other
    texture set, normal map only), 9 / 13 (plastic without thin film, linear-sRGB colour), 12 / 14 (four rough BK7 glass / coloured
    rough plastic heroes), 16 / 18 (near-smooth coat, FloatTexture coat thickness); glass and plastic with a FloatTexture roughness
    that switches between the specular and the microfacet branch across the surface (scene 27, not a reference scene); TWO environment
    lights whose radiance and MIS pdfs are summed (scene 29, scene.rs:185-231, mis_renderer.rs:205-214) and a textured emitter whose
    radiance is looked up at the hit / sampled uv and whose light-pick weight at uv (0.5, 0.5) (scene 30, emissive_material.rs:48-79)."""
    pair = {}
    for name, be in (("gpu", product), ("cpu", oracle)):
        sc = be.new_scene()
        pair[name] = (sc, pkg.scenes.load_scene(sc, scene_id, 128, 96, tex_size=256))
    oracle.set_faithful(pair["cpu"][0], False)
    rng = np.random.default_rng(scene_id)
    n = 30000
    xys = np.stack([rng.integers(0, 128, n), rng.integers(0, 96, n), rng.integers(0, 64, n)], 1).astype(np.uint32)
    prm = pkg.make_params(64, strategy, "sobol")
    Lg, lg, pg = pair["gpu"][0].probe_radiance(pair["gpu"][1], prm, xys)
    Lc, lc, pc = pair["cpu"][0].probe_radiance(pair["cpu"][1], prm, xys)
    assert np.array_equal(lg, lc)
    same_term = np.all(pg == pc, axis=1)
    assert same_term.mean() >= 0.995
    with np.errstate(invalid="ignore"):
        close = np.all((np.abs(Lg - Lc) <= 1e-3 * np.abs(Lc) + 1e-4) | (np.isnan(Lg) & np.isnan(Lc)), axis=1)   # (NaN samples: scene 11 under MIS, see below)
    if os.environ.get("MI355PT_PARITY_LOG"):                              # measurement run behind PER_SAMPLE_MIN (below)
        with open(os.environ["MI355PT_PARITY_LOG"], "a") as f:
            f.write(f'{{"scene": {scene_id}, "strategy": "{strategy}", "close": {close.mean():.6f}, "same_term": {same_term.mean():.6f}}}\n')
    assert close.mean() >= PER_SAMPLE_MIN, close.mean()
    # ... and two orders of magnitude tighter: the same paths, so the radiance agrees to the rounding of a few operations.  Measured
    # (tools/ulp_hist.py, profiles/r03_ulp_hist.jsonl): half of the samples bit-equal, 96 % within 1e-6, >= 99.95 % within 1e-5 (scene 19,
    # whose every sample reads the environment map through atan2f / acosf: 99.84 %; scene 29 with its two maps: 99.64 %)
    with np.errstate(invalid="ignore"):
        tight = np.all((np.abs(Lg - Lc) <= 1e-5 * np.abs(Lc) + 1e-12) | (np.isnan(Lg) & np.isnan(Lc)), axis=1)
    assert tight.mean() >= (0.995 if scene_id == 29 else 0.997), tight.mean()
    # The reference accumulates NaN samples without complaint (sensor.rs:42 only logs).  Rough SF11 glass under MIS makes
    # some: below 370 nm the eta LUT is 0 -> eta' = 1 (dielectric.rs:144-148), the "refracted" ray is -wo, and whenever
    # dot(wi,wm) + dot(wo,wm) rounds to exactly 0 the sample has f = 0 (spectrum / 0 -> 0), pdf = inf and the MIS weight
    # inf/(inf+0) = NaN.  Whether the sum rounds to 0 depends on the last ulp of the sampled normal: with the shading point rebuilt the
    # reference's way (round 3) both sides make the SAME NaN samples, all of them on hero wavelengths below 370 nm.
    nan_g, nan_c = np.isnan(Lg).any(axis=1), np.isnan(Lc).any(axis=1)
    bad = nan_g | nan_c
    assert bad.mean() <= 5e-3, bad.mean()
    assert np.count_nonzero(nan_g != nan_c) <= 1, (int(nan_g.sum()), int(nan_c.sum()))
    assert np.all(lg[nan_g, 0] < 370.0) and np.all(lc[nan_c, 0] < 370.0)
    assert abs(Lg[~bad, 0].mean() - Lc[~bad, 0].mean()) <= 2e-3 * Lc[~bad, 0].mean()
    # (the probe is the per-sample log of the scene's own kernel specialisation, written by the production launch)
    # compare a small image too (reference metric, regression_test.rs:6-40)
    prm8 = pkg.make_params(8, strategy, "sobol")
    qg = product.quantize_u8(product.render(pair["gpu"][0], pair["gpu"][1], prm8))
    qc = oracle.quantize_u8(oracle.render(pair["cpu"][0], pair["cpu"][1], prm8))
    assert linear_rmse_u8(qg, qc) <= 2e-3      # (the reference's own pass bar is 0.05; round 2 needed 0.01 ... 0.03 here)


@pytest.mark.parametrize("strategy", ["pt", "nee", "mis"])
def test_furnace_environment_light_gpu(product, pkg, strategy):
    """The furnace property of tests/test_oracle.py on the product (no oracle involved): albedo 0.5 under a constant sky."""
    from conftest import furnace_ratio
    sc = product.new_scene()
    cam = pkg.scenes.load_scene(sc, 23, 256, 192)
    r = furnace_ratio(product.render(sc, cam, pkg.make_params(256, strategy, "sobol")))
    assert np.all(np.abs(r - 0.5) <= 0.01), r


@pytest.mark.parametrize("scene_id", [24, 25])
def test_degenerate_bvh_scenes(product, oracle, pkg, scene_id):
    """A scene whose BVH root is a leaf (one triangle) or a single two-triangle leaf: the cooperative traversal starts in a leaf."""
    pair = {}
    for name, be in (("gpu", product), ("cpu", oracle)):
        sc = be.new_scene()
        pair[name] = (sc, pkg.scenes.load_scene(sc, scene_id, 64, 48))
    oracle.set_faithful(pair["cpu"][0], False)
    prm = pkg.make_params(16, "nee", "sobol")
    img_g = product.render(pair["gpu"][0], pair["gpu"][1], prm)
    img_c = oracle.render(pair["cpu"][0], pair["cpu"][1], prm)
    assert img_c.mean() > 0.01 and linear_rmse_u8(product.quantize_u8(img_g), oracle.quantize_u8(img_c)) <= 0.01
    o = np.zeros((4, 3), np.float32); o[:, 2] = 6.0; o[:, 1] = 1.5
    d = np.array([[0, 0, -1], [0.05, 0.1, -1], [0.9, 0, -0.1], [-0.2, 0.3, -1]], np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    tg = pair["gpu"][0].probe_intersect(o - np.array(pair["gpu"][1].position, np.float32), d)
    tc = pair["cpu"][0].probe_intersect(o - np.array(pair["cpu"][1].position, np.float32), d)
    assert np.array_equal(tg[0] > 0, tc[0] > 0) and np.allclose(tg[0], tc[0], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("strategy", ["pt", "nee", "mis"])
def test_scene_without_lights_is_black(product, pkg, strategy):
    sc = product.new_scene()
    cam = pkg.scenes.load_scene(sc, 26, 64, 48)
    img = product.render(sc, cam, pkg.make_params(8, strategy, "sobol"))
    assert np.array_equal(img, np.zeros_like(img))


def test_work_counters_match_oracle(product, oracle, pkg):
    """the per-sample work counts behind the algorithmic-bytes figure.  The instrumented kernel (collect_stats = 1: plain
    near-first traversal, what DESIGN.md calls the canonical counts) counts its own rays, hits, bounces, node and triangle steps; the
    oracle traces the same samples and, for every ray, ALSO walks the product's exported tree (mi355pt_scene_export_bvh -> FlatBvh) in
    that plain order with its own slab and triangle tests.  Everything must agree within 2 % (measured: < 0.1 %) — the node count is
    78 % of the bytes/sample numerator, so it is checked by an implementation that is not the one being measured."""
    import torch
    W, H, spp = 256, 192, 16
    pair = {}
    for name, be in (("gpu", product), ("cpu", oracle)):
        sc = be.new_scene()
        pair[name] = (sc, pkg.scenes.load_scene(sc, 3, W, H, tex_size=256))
    oracle.set_faithful(pair["cpu"][0], False)
    nodes, tris, root = product.export_bvh(pair["gpu"][0])
    assert nodes.shape[0] > 1000 and tris.shape[0] == 7202
    oracle.set_flat_bvh(pair["cpu"][0], nodes, tris, root)
    st = pkg.ffi.Stats()
    a = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    product.render_accum_device(pair["gpu"][0], pair["gpu"][1], pkg.make_params(spp, "mis", "sobol", collect_stats=1), 0, spp, a.data_ptr(), None, stats=st)
    g = st.as_dict()
    oracle.counters(pair["cpu"][0], reset=True)
    oracle.render_accum(pair["cpu"][0], pair["cpu"][1], pkg.make_params(spp, "mis", "sobol"), 0, spp, threads=8, counters=True)
    c = oracle.counters(pair["cpu"][0])
    assert g["samples"] == c["samples"] == W * H * spp
    for k in ("closest_rays", "shadow_rays", "closest_hits", "bounces"):
        assert abs(g[k] - c[k]) <= 0.02 * c[k], (k, g[k], c[k])
    for kg, kc in (("nodes_closest", "flat_closest_nodes"), ("tris_closest", "flat_closest_tris"), ("nodes_shadow", "flat_any_nodes"),
                   ("tris_shadow", "flat_any_tris")):
        assert abs(g[kg] - c[kc]) <= 0.02 * c[kc], (kg, g[kg], c[kc])


def test_full_size_properties(product, pkg):
    """BASELINE configs[1] frame size (1920x1080): size-independent properties of the film path — the eight tile shards of an
    8-GPU job sum to the single-GPU film, sample ranges compose (indices [0,4) + [4,8) = [0,8) up to float summation order),
    and the resolve is the documented per-pixel function of the sums."""
    import torch
    sc = product.new_scene()
    W, H = 1920, 1080
    cam = pkg.scenes.load_scene(sc, 3, W, H, tex_size=256)
    spp = 1024

    def accum(s0, s1, shard=0, shards=1):
        a = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        product.render_accum_device(sc, cam, pkg.make_params(spp, "mis", "sobol", shard_index=shard, shard_count=shards), s0, s1, a.data_ptr(), None)
        torch.cuda.synchronize()
        return a
    full = accum(0, 8)
    parts = sum(accum(0, 8, k, 8) for k in range(8))
    assert torch.equal(parts, full)                                        # disjoint tiles: exact
    assert bool((full.sum(dim=2) > 0).float().mean() > 0.95)               # (spectral -> RGB sums can be slightly negative)
    two = accum(0, 4) + accum(4, 8)
    assert float((two - full).abs().max()) <= 1e-4 * float(full.abs().max())
    out = torch.empty_like(full)
    product.film_resolve_device(full.data_ptr(), W * H, 8, out.data_ptr(), None)
    torch.cuda.synchronize()
    x = torch.clamp(full / 8.0, min=0.0)
    y = x / (1.0 + x)
    ref = torch.where(y <= 0.0031308, 12.92 * y, 1.055 * y.pow(1.0 / 2.4) - 0.055)
    assert float((out - ref).abs().max()) <= 2e-6


@pytest.mark.parametrize("w,h,spp,max_depth,strategy,sampler", [
    (37, 23, 8, 16, "mis", "sobol"),      # ragged: neither side a multiple of the 8x8 tile, odd log2(spp)
    (1, 1, 64, 16, "mis", "sobol"),       # a single pixel
    (9, 130, 2, 16, "nee", "sobol"),      # tall and thin, log2(spp) = 1
    (64, 48, 1, 16, "pt", "sobol"),       # one sample per pixel
    (50, 40, 12, 16, "mis", "sobol"),     # spp not a power of two
    (48, 32, 16, 1, "mis", "sobol"),      # max_depth 1: direct light only
    (33, 17, 32, 3, "nee", "random"),     # counter-hash sampler on a ragged frame
])
def test_edge_case_frames_match_oracle(product, oracle, pkg, w, h, spp, max_depth, strategy, sampler):
    """Ragged / tiny frames, odd sample counts and depth limits through the render path (tile masking, chunked sample ranges
    with film atomics, Sobol tables at small digit counts): same image as the oracle (reference metric, RMSE of the 8-bit frames)."""
    pair = {}
    for name, be in (("gpu", product), ("cpu", oracle)):
        sc = be.new_scene()
        pair[name] = (sc, pkg.scenes.load_scene(sc, 0, w, h))
    oracle.set_faithful(pair["cpu"][0], False)
    prm = pkg.make_params(spp, strategy, sampler, max_depth=max_depth)
    img_g = product.render(pair["gpu"][0], pair["gpu"][1], prm)
    img_c = oracle.render(pair["cpu"][0], pair["cpu"][1], prm)
    assert img_g.shape == (h, w, 3) and np.isfinite(img_g).all()
    assert linear_rmse_u8(product.quantize_u8(img_g), oracle.quantize_u8(img_c)) <= 0.012
    assert abs(float(img_g.mean()) - float(img_c.mean())) <= 0.01 + 0.02 * float(img_c.mean())


def test_shards_tile_the_frame(product, pkg):
    """Multi-GPU decomposition: rendering shard k of N touches only its tiles and the shards sum to the full frame."""
    import ctypes as C
    sc = product.new_scene()
    cam = pkg.scenes.load_scene(sc, 0, 96, 64)
    full = product.render(sc, cam, pkg.make_params(8, "mis", "sobol"))
    parts = [product.render(sc, cam, pkg.make_params(8, "mis", "sobol", shard_index=k, shard_count=3)) for k in range(3)]
    # untouched pixels resolve to exactly 0
    nz = [(p != 0).any(axis=2) for p in parts]
    assert not (nz[0] & nz[1]).any() and not (nz[0] & nz[2]).any() and not (nz[1] & nz[2]).any()
    assert np.array_equal(sum(parts), full)


@pytest.mark.parametrize("scene_id,w,h,spp", [(3, 160, 104, 1024), (8, 100, 70, 256), (0, 64, 48, 4096), (0, 40, 24, 16384),
                                               (0, 8, 16384, 4096),    # 38-bit sample indices: the prefix tables force a 2x2 block
                                               (3, 96, 64, 512), (0, 64, 48, 2048)])   # odd log2(spp): the half digit at the bottom of the index
def test_launch_shape_does_not_change_the_frame(product, oracle, pkg, scene_id, w, h, spp):
    """The launcher picks the work-item shape from the number of sample indices per launch (8x8 tiles for short launches,
    4x4 / 2x2 / 1x1 pixel blocks for longer ones, api.cpp): the film of the whole job in ONE launch must equal the film
    accumulated over launches of 64 (and of 16) sample indices up to float summation order, shards included, and must match
    the oracle's frame like any other."""
    import torch
    sc = product.new_scene()
    cam = pkg.scenes.load_scene(sc, scene_id, w, h, tex_size=64)
    prm = pkg.make_params(spp, "mis", "sobol")
    films = {}
    # (16 384 spp in launches of 4 096: single-pixel items whose top sample digit is part of the Sobol prefix tables)
    for name, step in (("one", spp), ("by64", 64), ("by16", 16 if spp <= 4096 else 4096)):
        a = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
        for s0 in range(0, spp, step):
            product.render_accum_device(sc, cam, prm, s0, s0 + step, a.data_ptr(), None)
        torch.cuda.synchronize()
        films[name] = a.cpu().numpy()
    sharded = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    for k in range(3):
        product.render_accum_device(sc, cam, pkg.make_params(spp, "mis", "sobol", shard_index=k, shard_count=3), 0, spp, sharded.data_ptr(), None)
    torch.cuda.synchronize()
    ref = films["one"]
    assert ref.mean() > 0.01 * spp * 0.1
    for name in ("by64", "by16"):
        np.testing.assert_allclose(films[name], ref, rtol=2e-4, atol=1e-4 * spp * 0.01)
    # (a third of the tiles per launch: the launcher may split sample ranges differently, so again only the summation order moves)
    np.testing.assert_allclose(sharded.cpu().numpy(), ref, rtol=2e-4, atol=1e-4 * spp * 0.01)
    if spp <= 1024:
        so = oracle.new_scene()
        cam_o = pkg.scenes.load_scene(so, scene_id, w, h, tex_size=64)
        oracle.set_faithful(so, False)
        img_o = oracle.render(so, cam_o, prm)
        img_g = product.render(sc, cam, prm)
        # (solid glass: a refracted path that flips at an edge or at a Russian-roulette threshold moves its pixel by a lot; the
        # reference's own regression thresholds are 0.05-0.085, regression_test.rs:109-659)
        assert linear_rmse_u8(product.quantize_u8(img_g), oracle.quantize_u8(img_o)) <= 2e-3


# (name, scene, W, H, spp, strategy, shard_count): every BASELINE.json config at its TRUE size; the shard holds 4-9 tiles spread over the frame
BASELINE_CONFIGS = [("C2", 3, 1920, 1080, 1024, "mis", 4001), ("C3", 10, 1920, 1080, 4096, "mis", 4001),
                    ("C4", 8, 4096, 4096, 1024, "mis", 30011), ("C5", 17, 1920, 1080, 16384, "nee", 8009)]


@pytest.mark.parametrize("name,scene_id,w,h,spp,strategy,shard_count", BASELINE_CONFIGS, ids=[c[0] for c in BASELINE_CONFIGS])
def test_baseline_configs_at_true_size_match_the_oracle(product, oracle, pkg, name, scene_id, w, h, spp, strategy, shard_count):
    """The RENDER kernel against the oracle at each BASELINE config's true resolution and spp, where C3 / C4 / C5's u32 Morton index
    truncates (z_sobol_sampler.rs:198-201: 2 log2(res) + log2(spp) = 34 / 34 / 36 bits) and distant pixels share sample sequences.
    One sparse shard (the tile through the frame centre, i.e. on the hero, and every shard_count-th tile from there: tiles beyond x or
    y = 1024 / 2048, where dropped Morton bits matter) is rendered over the WHOLE sample range by the production launch path — the
    2x2-block / single-pixel work items, the LDS Sobol prefix tables incl. sample_prefix_digits, and for C5 the split into launches of
    4 096 sample indices — with the per-sample log on, and by the oracle's render_accum on the same shard.  Film: resolved shard pixels
    to the frame-test bar.  Samples: a seeded subset of the log against the oracle's per-sample radiance, wavelengths bit-equal."""
    tiles_x, tiles_y = (w + 7) // 8, (h + 7) // 8
    centre = (tiles_y // 2) * tiles_x + tiles_x // 2
    shard_index = centre % shard_count
    tiles = np.arange(shard_index, tiles_x * tiles_y, shard_count)
    tx, ty = tiles % tiles_x, tiles // tiles_x
    assert 4 <= len(tiles) <= 9 and centre in tiles
    assert (tx * 8 >= w // 2).any() and (ty * 8 >= h // 2).any()          # pixels whose high Morton bits are set
    pair = {}
    for be_name, be in (("gpu", product), ("cpu", oracle)):
        sc = be.new_scene()
        # C2 with bench.py's own input: the default 1024 x 1024 albedo and normal textures of scene 3 (the other configs have no textures)
        pair[be_name] = (sc, pkg.scenes.load_scene(sc, scene_id, w, h, tex_size=1024 if name == "C2" else 256))
    oracle.set_faithful(pair["cpu"][0], False)
    prm = pkg.make_params(spp, strategy, "sobol", shard_index=shard_index, shard_count=shard_count)
    L, lam, pdf, film_g = product.render_sample_log(pair["gpu"][0], pair["gpu"][1], prm, 0, spp, want_accum=True)
    assert L.shape == (len(tiles), 64, spp, 4)
    film_c, _ = oracle.render_accum(pair["cpu"][0], pair["cpu"][1], prm, 0, spp, threads=min(os.cpu_count() or 1, 16))
    mask = np.zeros((h, w), bool)
    for x0, y0 in zip(tx * 8, ty * 8):
        mask[y0:y0 + 8, x0:x0 + 8] = True
    assert not film_g[~mask].any() and not film_c[~mask].any()             # nothing outside the shard's tiles
    rg, rc = oracle.film_resolve(film_g[mask], spp), oracle.film_resolve(film_c[mask], spp)
    rmse = float(np.sqrt(np.mean((rg - rc) ** 2)))
    off = int((np.abs(rg - rc).max(axis=1) > 0.01).sum())
    if os.environ.get("MI355PT_FRAME_LOG"):
        with open(os.environ["MI355PT_FRAME_LOG"], "a") as f:
            f.write(f'{{"config": "{name}", "true_size_shard_rmse": {rmse:.3e}, "off": {off}}}\n')
    assert film_c[mask].mean() > 0.01 * spp and rmse <= 1.5e-4 and off == 0, (rmse, off)      # (round 2: 5e-4 and <= 2 pixels)
    # the log's film is the film: summing the logged samples' sensor responses is what add_sample did (spot check through the oracle's resolve
    # is not possible per sample, so compare per-sample radiance instead)
    rng = np.random.default_rng(spp + scene_id)
    n = 6000
    k, pix, s = rng.integers(0, len(tiles), n), rng.integers(0, 64, n), rng.integers(0, spp, n)
    s[:64] = spp - 1 - np.arange(64) % 8                                   # the top of the sample range (last launch of a split job)
    xs, ys = tx[k] * 8 + pix % 8, ty[k] * 8 + pix // 8
    keep = (xs < w) & (ys < h)
    k, pix, s, xs, ys = k[keep], pix[keep], s[keep], xs[keep], ys[keep]
    xys = np.stack([xs, ys, s], 1).astype(np.uint32)
    Lc, lc, pc = pair["cpu"][0].probe_radiance(pair["cpu"][1], pkg.make_params(spp, strategy, "sobol"), xys)
    Lg, lg, pg = L[k, pix, s], lam[k, pix, s], pdf[k, pix, s]
    assert np.array_equal(lg, lc) and np.array_equal(pg, pc)               # wavelengths and termination: straight from the Sobol bits
    close = np.all(np.abs(Lg - Lc) <= 1e-3 * np.abs(Lc) + 1e-4, axis=1)
    assert close.mean() >= 0.9995, close.mean()


@pytest.mark.parametrize("device_ids", [[0], [0, 0, 0], "all", "all_reversed"])
def test_render_multi_is_render(product, pkg, device_ids):
    """mi355pt_scene_build_multi + mi355pt_render_multi (one process, the frame's tiles dealt to several devices, each peer's shard
    packed to a compact film and pushed to the first device, unpacked there, resolved) return the frame mi355pt_render returns.  On the
    one-GPU box the device list repeats device 0, which exercises the replicas, the per-device streams and events, the sharding, the
    pack / peer copy / unpack and the resolve; with one device the frame is bit-identical, with several shards only the float summation
    order inside a pixel may move (the launcher may split sample ranges differently for a share of the tiles).  "all": every GPU the box
    has, as real peers over xGMI (skipped with one GPU) — also with the LAST device gathering, so that the first device of the list is
    not the process's current one and the per-scene launch context has to follow the scene."""
    import torch
    if isinstance(device_ids, str):
        if torch.cuda.device_count() < 2:
            pytest.skip("one GPU: real peers need a multi-GPU box")
        device_ids = list(range(torch.cuda.device_count()))[::-1 if device_ids == "all_reversed" else 1]
    assert all(d < torch.cuda.device_count() for d in device_ids)
    ref_sc = product.new_scene()
    cam = pkg.scenes.load_scene(ref_sc, 3, 200, 150, tex_size=128)
    prm = pkg.make_params(64, "mis", "sobol")
    ref = product.render(ref_sc, cam, prm)
    sc = product.new_scene()
    cam2 = pkg.scenes.load_scene(sc, 3, 200, 150, tex_size=128, build=False)
    product.build_multi(sc, cam2, device_ids)
    for _ in range(2):                                                      # the second call reuses the per-device films and streams
        img = product.render_multi(sc, cam2, prm)
        if len(device_ids) == 1:
            assert np.array_equal(img, ref)
        else:
            assert float(np.abs(img - ref).max()) <= 2e-5
    with pytest.raises(RuntimeError):                                       # a multi-device scene shards the frame itself
        product.render_multi(sc, cam2, pkg.make_params(64, "mis", "sobol", shard_index=0, shard_count=2))
    with pytest.raises(RuntimeError):                                       # a single-device scene is not a multi-device scene
        product.render_multi(ref_sc, cam, prm)


@pytest.mark.parametrize("scene_id,strategy", [(3, "mis"), (17, "nee"), (19, "mis"), (8, "pt")])
def test_frames_are_bit_identical_from_run_to_run(product, pkg, scene_id, strategy):
    """The wave state machine (path hand-out, cooperative traversals with subtree stealing, LDS film tile, chunk combine) has no float
    atomics on the film and no order that depends on timing: two renders of the same job give the same bits.  (DESIGN.md 5.0 leans on this when it prices a material sort through L2 queues.)"""
    sc = product.new_scene()
    cam = pkg.scenes.load_scene(sc, scene_id, 256, 192, tex_size=128)
    prm = pkg.make_params(64, strategy, "sobol")
    a = product.render(sc, cam, prm)
    b = product.render(sc, cam, prm)
    assert np.array_equal(a, b)
    assert np.isfinite(a).all() and float(a.mean()) > 0.0


def test_dielectric_roughness_map_is_used(product, oracle, pkg):
    """Scene 27 (glass + plastic with a FloatTexture roughness) against scene 28 (same heroes, constant roughness 0): the map must
    change the frame on both sides, and by the same amount."""
    frames = {}
    for name, be in (("gpu", product), ("cpu", oracle)):
        for sid in (27, 28):
            sc = be.new_scene()
            cam = pkg.scenes.load_scene(sc, sid, 128, 96, tex_size=256)
            if name == "cpu":
                oracle.set_faithful(sc, False)
            frames[name, sid] = be.quantize_u8(be.render(sc, cam, pkg.make_params(64, "mis", "sobol")))
    d_gpu = linear_rmse_u8(frames["gpu", 27], frames["gpu", 28])
    d_cpu = linear_rmse_u8(frames["cpu", 27], frames["cpu", 28])
    assert d_gpu > 0.01 and abs(d_gpu - d_cpu) <= 0.2 * d_cpu, (d_gpu, d_cpu)
    assert linear_rmse_u8(frames["gpu", 27], frames["cpu", 27]) <= 2e-3


def test_cpp_host_cli_matches_python_binding(product, pkg, tmp_path):
    """The C++ mirror of the reference's renderer API (toy-cpu-pathtracing_amd/host: Scene::load_obj, load_scene_3,
    RendererImage::render/save, main.rs flags) must produce the picture the ctypes path produces: same library, same
    assets through OBJ/PPM files instead of arrays."""
    import subprocess, sys
    from PIL import Image
    root = pkg.ffi.ROOT
    exe = os.path.join(root, "toy-cpu-pathtracing_amd", "host", "mi355pt")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    assets_dir = str(tmp_path / "assets")
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "export_assets.py"), assets_dir])
    for scene_id, renderer in ((3, "mis"), (17, "nee"), (7, "mis"), (1, "nee"), (19, "mis"), (13, "mis"), (15, "mis"), (18, "nee"), (12, "mis")):
        out = str(tmp_path / f"cli_{scene_id}.png")
        env = dict(os.environ, MI355PT_ASSETS=assets_dir, MI355PT_DATA=os.path.join(root, "toy-cpu-pathtracing_amd", "data"))
        r = subprocess.run([exe, "--scene", str(scene_id), "--renderer", renderer, "--sampler", "sobol", "--spp", "8", "--width", "96",
                            "--height", "64", "--output", out], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "Finish rendering" in r.stdout
        cli = np.asarray(Image.open(out).convert("RGB"))
        sc = product.new_scene()
        cam = pkg.scenes.load_scene(sc, scene_id, 96, 64)
        ref = product.quantize_u8(product.render(sc, cam, pkg.make_params(8, renderer, "sobol")))
        assert cli.shape == ref.shape
        assert (np.abs(cli.astype(int) - ref.astype(int)) <= 1).mean() >= 0.999
        assert linear_rmse_u8(cli, ref) <= 1e-3
    # the CLI's extension flags: --albedo-lut is mi355pt_params.albedo_lut, --gpus 1 is the single-device path
    out = str(tmp_path / "cli_lut.png")
    r = subprocess.run([exe, "--scene", "17", "--renderer", "nee", "--sampler", "sobol", "--spp", "8", "--width", "96", "--height", "64",
                        "--albedo-lut", "--gpus", "1", "--output", out], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cli = np.asarray(Image.open(out).convert("RGB"))
    sc = product.new_scene()
    cam = pkg.scenes.load_scene(sc, 17, 96, 64)
    ref = product.quantize_u8(product.render(sc, cam, pkg.make_params(8, "nee", "sobol", albedo_lut=1)))
    assert (np.abs(cli.astype(int) - ref.astype(int)) <= 1).mean() >= 0.999


def test_regression_runner_rehearsal(product, pkg, tmp_path):
    """The 42-case runner end to end on one case: the CLI's own frame stands in for the (absent) golden PNG, so the RMSE must be 0 and the
    case must pass its threshold; without --rehearsal the same file is refused by its sha256."""
    import subprocess, sys
    root = pkg.ffi.ROOT
    exe = os.path.join(root, "toy-cpu-pathtracing_amd", "host", "mi355pt")
    assets_dir = str(tmp_path / "assets")
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "export_assets.py"), assets_dir])
    refs = tmp_path / "refs"; refs.mkdir()
    env = dict(os.environ, MI355PT_ASSETS=assets_dir, MI355PT_DATA=os.path.join(root, "toy-cpu-pathtracing_amd", "data"))
    subprocess.check_call([exe, "--scene", "0", "--renderer", "pt", "--sampler", "sobol", "--spp", "512", "--width", "200", "--height", "150",
                           "--output", str(refs / "reference_pt_sobol.png")], env=env, stdout=subprocess.DEVNULL)
    runner = os.path.join(root, "tools", "run_reference_regressions.py")
    r = subprocess.run([sys.executable, runner, "--references", str(refs), "--assets", assets_dir, "--only", "reference_pt_sobol", "--rehearsal"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "PASS" in r.stdout and "RMSE 0.000000" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, runner, "--references", str(refs), "--only", "reference_pt_sobol"], capture_output=True, text=True)
    assert r.returncode == 1 and "CHECKSUM" in r.stdout


def test_bench_n_gpus_reproduces_the_one_gpu_film_bit_for_bit():
    """bench.py --gpus 2 (one rank per GPU over RCCL: tiles dealt round-robin, one film reduce) against bench.py --gpus 1 on a small
    frame: the digest of the reduced film printed in the JSON line (`film_check.sha256`) is the same — disjoint tiles and a deterministic
    in-wave schedule make the N-GPU film the 1-GPU film, bit for bit.  Needs two GPUs (skipped on the one-GPU box); both runs are child
    processes, so this process never initialises a second device."""
    import json
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU: the RCCL path needs a multi-GPU box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1", "--width", "256", "--height", "200",
                            "--spp", "64", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1]
        j = json.loads(line)
        assert j["n_gpus"] == n and j["film_check"] and j["film_check"]["mean"] > 0.01
        digests[n] = j["film_check"]["sha256"]
    assert digests[1] == digests[2], digests


def test_rr_gate_slack_is_refused_unless_unlocked(product, pkg):
    """mi355pt_params.rr_gate_slack changes what a render call computes; a drop-in caller's uninitialised struct must not be able to do
    that by accident: the library refuses a non-zero value unless mi355pt_debug_unlock(1) (include/mi355pt_debug.h) was called."""
    sc = product.new_scene()
    cam = pkg.scenes.load_scene(sc, 0, 32, 24)
    prm = pkg.make_params(4, "mis", "sobol", rr_gate_slack=1e-5)
    was = product.debug_unlock(False)
    try:
        with pytest.raises(RuntimeError, match="rr_gate_slack must be 0"):
            product.render(sc, cam, prm)
        product.render(sc, cam, pkg.make_params(4, "mis", "sobol"))          # the reference's gate needs no switch
    finally:
        product.debug_unlock(was)
    product.render(sc, cam, prm)


def test_device_sincos_equals_the_host_libm(product):
    """The reference's sin / cos are the host libm's (Rust f32::sin_cos -> sinf / cosf; GGX normal sampling, cosine hemisphere, conductor
    Fresnel, environment map).  csrc/pt_libm.hpp restates glibc's algorithm for the device; mi355pt_probe_sincos runs the render kernels'
    function (ref_sincosf) on the GPU and compares with THIS host's libm bit for bit: all 1 087 M floats of [0, 2 pi) would take a minute of host libm
    time, so every 64th of them (17 M), every 4 099th float of (-120, 120) and the first 2^20 floats above 0 (subnormals, tiny arguments).
    Beyond +-120 the kernels fall back to the device libm (no call site produces such angles): a range that must merely stay finite."""
    two_pi_bits = 0x40C90FDB
    for first, stride, n in ((0, 64, two_pi_bits // 64), (0, 4099, 0x42F00000 // 4099), (0x80000000, 4099, 0x42F00000 // 4099), (0, 1, 1 << 20)):
        compared, bad_s, bad_c = product.probe_sincos(first, stride, n)
        assert compared == n and bad_s == 0 and bad_c == 0, (hex(first), stride, n, bad_s, bad_c)


@pytest.mark.parametrize("x,y,s", [(1895, 369, 1865), (1085, 615, 2904)])
def test_edge_on_thin_film_sample_stays_finite_like_the_reference(product, oracle, pkg, x, y, s):
    """Two samples of BASELINE configs[2] (scene 10, 1920x1080, 4096 spp, MIS) at its TRUE size that round 3's first exact-hit build turned
    into NaN pixels of the film (found by bench.py's film digest: mean = nan): a late vertex hits the thin-film hero edge-on (wo.z = -7e-9 in
    the exact shading frame), the specular sample has f = 0 and pdf = NaN on both sides, and the next vertex is not a light.  The reference
    adds T * SampledSpectrum::zero() there (base_renderer.rs:124-131) — an exact zero — and ends the path at the roulette (max of NaNs is
    -inf, u < -inf fails); the kernel used to add T * ((f * 0) * (1 / pdf)) = NaN.  The whole tile of each sample over a window of sample
    indices around it, production launch, against the oracle: finite, and the same radiance."""
    W, H, S = 1920, 1080, 4096
    sc, so = product.new_scene(), oracle.new_scene()
    cam = pkg.scenes.load_scene(sc, 10, W, H, tex_size=64)
    cam_o = pkg.scenes.load_scene(so, 10, W, H, tex_size=64)
    oracle.set_faithful(so, False)
    tiles_x = (W + 7) // 8
    prm = pkg.make_params(S, "mis", "sobol", shard_index=(y // 8) * tiles_x + x // 8, shard_count=tiles_x * ((H + 7) // 8))
    s0 = s - 8
    L, lam, pdf = product.render_sample_log(sc, cam, prm, s0, s0 + 16)
    assert L.shape[0] == 1 and np.isfinite(L).all()
    pix = (y & 7) * 8 + (x & 7)
    xys = np.stack([np.full(16, x), np.full(16, y), np.arange(s0, s0 + 16)], 1).astype(np.uint32)
    Lc, lc, pc = so.probe_radiance(cam_o, pkg.make_params(S, "mis", "sobol"), xys)
    assert np.array_equal(lam[0, pix], lc) and np.array_equal(pdf[0, pix], pc)
    assert np.all(np.abs(L[0, pix] - Lc) <= 1e-3 * np.abs(Lc) + 1e-6)


@pytest.mark.parametrize("scene_id,strategy", [(3, "mis"), (4, "mis"), (8, "nee")])
def test_lowering_paths_agree_with_the_oracle(product, oracle, pkg, scene_id, strategy):
    """The three ways the product rebuilds the reference's local-space hit (csrc/scene.cpp picks one per scene / instance;
    mi355pt_scene_debug_set_lowering forces the fallbacks):
      local     every instance is the same pure translation: the triangle array holds local vertices, every triangle test of every traversal
                runs on the reference's local ray (scene_info: tri_space=local) — what the Cornell-class BASELINE scenes take;
      identity  render-space traversal, the triangle found re-tested in local space, translation instances skip the 3x3 products;
      general   the same with every instance through the full matrix path — what scaled / rotated heroes take.
    All three must trace the oracle's paths (frame bar), and identity == general bit for bit (1 * a + 0 * b + 0 * c is a)."""
    frames = {}
    for mode, lowering in (("local", "auto"), ("identity", "no_local_tris"), ("general", "general")):
        sc = product.new_scene()
        sc.debug_set_lowering(lowering)
        cam = pkg.scenes.load_scene(sc, scene_id, 64, 48, tex_size=128)
        assert ("tri_space=local" in product.scene_info(sc)) == (mode == "local")
        frames[mode] = product.render(sc, cam, pkg.make_params(64, strategy, "sobol"))
    so = oracle.new_scene()
    cam_o = pkg.scenes.load_scene(so, scene_id, 64, 48, tex_size=128)
    oracle.set_faithful(so, False)
    c = oracle.render(so, cam_o, pkg.make_params(64, strategy, "sobol"))
    assert np.array_equal(frames["identity"], frames["general"])
    for mode, g in frames.items():
        rmse = float(np.sqrt(np.mean((g - c) ** 2)))
        off = int((np.abs(g - c).max(axis=2) > 0.01).sum())
        assert rmse <= 5e-4 and off <= 2, (mode, rmse, off)      # (the render-space any-hit of the two fallback paths leaves one sample of scene 4: 3.1e-4 / 1 pixel)
    rmse = float(np.sqrt(np.mean((frames["local"] - c) ** 2)))
    assert rmse <= FRAME_BAR[0] and int((np.abs(frames["local"] - c).max(axis=2) > 0.01).sum()) == 0


def test_degenerate_triangles_stay_out_of_the_tree(product, oracle, pkg):
    """math::intersect_triangle rejects a triangle whose cross product is exactly zero before anything else (ray.rs:49-56): it can never be
    hit.  The product decides that once per triangle on the host (csrc/scene.cpp, on the vertices the traversal will test, with the
    device's arithmetic), leaves such triangles out of the tree and runs the traversals' triangle test without the check (+1.5 ... 5 %).
    Scene 33 = scene 25 with a collinear triple and a repeated vertex mixed into its first mesh: two triangles excluded, the frame scene
    25's bit for bit in every lowering path, and the oracle's (which tests and rejects them like the reference) within the frame bar."""
    frames = {}
    for lowering in ("auto", "general"):
        for sid in (25, 33):
            sc = product.new_scene()
            sc.debug_set_lowering(lowering)
            cam = pkg.scenes.load_scene(sc, sid, 64, 48)
            info = product.scene_info(sc)
            assert f"degenerate={2 if sid == 33 else 0} " in info and " tris=2 " in info, info
            frames[lowering, sid] = product.render(sc, cam, pkg.make_params(64, "nee", "sobol"))
        assert np.array_equal(frames[lowering, 25], frames[lowering, 33])
    so = oracle.new_scene()
    cam_o = pkg.scenes.load_scene(so, 33, 64, 48)
    oracle.set_faithful(so, False)
    c = oracle.render(so, cam_o, pkg.make_params(64, "nee", "sobol"))
    g = frames["auto", 33]
    assert c.mean() > 0.01 and float(np.sqrt(np.mean((g - c) ** 2))) <= FRAME_BAR[0] and int((np.abs(g - c).max(axis=2) > 0.01).sum()) == 0
