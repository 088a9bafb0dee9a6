#!/usr/bin/env python3
"""bench.py — Msamples/s of the spectral path-tracing sample loop on MI355X.

Workload (BASELINE.json configs[1]): scene3 (textured + normal-mapped Lambert hero in the Cornell room, synthetic
stand-in assets), MIS + ZSobol, 1920x1080, target 1024 spp.  One *step* = one pass of the hot path over one batch =
what one `RendererImage::render()` call of the reference does: the whole 1920x1080 frame at all 1024 sample indices of
the job, into a zeroed film (132.7 M pixel-samples x 16; `--spp-per-step S` < spp splits the job into consecutive
ranges of S sample indices per step instead, e.g. progressive refinement; jobs above 4096 spp use ranges of 4096).
Inputs (BVH, triangles, materials, LUTs, tables, textures) are resident in HBM before the timed region; the film
accumulators stay in HBM.

N > 1 (one rank per GPU; either launched by torch.distributed.run, or plainly as `python bench.py --gpus N`, in which
case this process starts torch.distributed.run itself as a CHILD before touching any GPU and relays rank 0's JSON
line): the frame's 8x8 pixel tiles are dealt round-robin to
the ranks (scene replicated); each rank renders its own tiles into its own linear film, with no data-path collective,
and when a frame's last sample index is done the films are summed onto rank 0 with ONE RCCL reduce over xGMI, inside
the timed region (strong scaling: the frame is fixed).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# VALU issue peak (same guide): 256 CUs x 4 SIMD-32 per CU, a wave64 VALU instruction occupies its SIMD for 2 cycles, 2.4 GHz max clock
PEAK_VALU_GINSTR = 256 * 4 * 2.4 / 2.0   # = 1228.8 G wave-level VALU instructions per second


def algorithmic_bytes_per_sample(st, spp_total):
    """SURVEY.md §8(d)'s byte model on the PRODUCTION traversal's own counts and record sizes (instrumented kernel, collect_stats = 2:
    the merged wave-cooperative walk of the 4-wide tree): 112 B read per DevNode4 visit (six plane quads + the links),
    48 B per triangle test (DevTri), per closest hit the winner's vertices again (48 B) + its DevTriShade record (112 B) + the material
    record (176 B), 16 B per spectrum evaluation, 48 B of CMF per sample, 120 B per textured lookup (4 texels + 8 rgb2spec cells
    are 144 B; the survey's figure is kept), film 12 B/pixel once.  Path state never leaves registers (0 B).  These are the bytes the
    kernel REQUESTS from its caches — the working set is L2-resident, so the figure is set against the L2 request rate the counters
    show, not against HBM (round 2 divided the reference-order BVH2 counts by the HBM peak and got 1.09: not a roofline fraction)."""
    n = max(st["samples"], 1)
    b = ((st["nodes_closest"] + st["nodes_shadow"]) * 112 + (st["tris_closest"] + st["tris_shadow"]) * 48 +
         st["closest_hits"] * (48 + 112 + 176) + st["spectrum_evals"] * 16 + st["textured_lookups"] * 120) / n
    return b + 48 + 12.0 / spp_total


def film_digest(t):
    """sha256 of the linear film (bit pattern, row-major H x W x 3 f32) + order-independent moments of it"""
    import hashlib
    a = t.detach().cpu().contiguous().numpy()
    import numpy as np
    fin = np.isfinite(a)                                    # (the reference accumulates NaN samples, sensor.rs:42: moments over the finite values, their count beside them)
    a64 = a.astype("float64")[fin]
    return hashlib.sha256(a.tobytes()).hexdigest(), float(a64.mean()) if a64.size else 0.0, float(a64.max()) if a64.size else 0.0, int((~fin).sum())


def find_profile(name, wl):
    """the committed counter pass (tools/profile_bench.sh -> profiles/<name>*.json) taken on exactly this workload"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", name + "*.json"))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("workload") == wl:
            return j, os.path.relpath(f, ROOT)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=0, help="sample indices per step; 0 = the whole job, at most 4096")
    ap.add_argument("--scene", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024, help="spp of the whole job (fixes the Sobol sequence)")
    ap.add_argument("--strategy", default="mis")
    ap.add_argument("--sampler", default="sobol")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU-baseline duration")
    ap.add_argument("--write-film-checksum", action="store_true", help="record this run's film digest in profiles/film_checksums.json (commit it)")
    args = ap.parse_args()


    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launched plainly: start one rank per GPU as children of this process (which has not touched a GPU and never will),
        # relay their output (rank 0 prints the JSON line) and exit with their code
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    import torch
    import torch.distributed as dist

    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # MI355PT_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, the film
    # reduce is staged through the host); the measured configuration is always nccl (= RCCL) with one GPU per rank
    backend = os.environ.get("MI355PT_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)

    pkg = importlib.import_module("toy-cpu-pathtracing_amd")
    mg = importlib.import_module("toy-cpu-pathtracing_amd.multigpu")
    prod = pkg.Product()
    scene = prod.new_scene()
    cam = pkg.scenes.load_scene(scene, args.scene, args.width, args.height)     # BVH build + upload to this rank's GPU
    W, H = args.width, args.height
    spp_job = args.spp
    sps = args.spp_per_step if args.spp_per_step > 0 else min(spp_job, 4096)
    n_slices = max(spp_job // sps, 1)
    frame_per_step = n_slices == 1      # every step is a complete frame: zeroed film in, reduced film out

    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    def render_slice(k):
        def render_accum(acc, shard_index, shard_count):
            p = pkg.make_params(spp_job, args.strategy, args.sampler, shard_index=shard_index, shard_count=shard_count)
            prod.render_accum_device(scene, cam, p, k * sps, (k + 1) * sps, acc.data_ptr(), stream)
        return render_accum

    def step(i, events=None):
        # this rank's tiles for the step's sample indices, added into the rank-local linear film (no collective)
        if frame_per_step:
            accum.zero_()
        if events:
            events[0].record()
        mg.render_frame_sharded(render_slice(i % n_slices), accum, rank, world, reduce=False)
        if events:
            events[1].record()       # brackets exactly the path-tracing launch on the stream it was launched on
        if frame_per_step:
            mg.reduce_film(accum, world)     # the frame's single film reduce (linear sums, pre-tonemap)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
        accum.zero_()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, ev[i])
    if not frame_per_step:
        mg.reduce_film(accum, world)     # the job's single film reduce (linear sums, pre-tonemap), timed
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    dt = mg.max_over_ranks(dt, world, "cuda")

    samples_per_step = W * H * sps
    total = samples_per_step * args.steps
    value = total / dt / 1e6

    # ---- what the timed launches rendered: the last step's film (reduced onto rank 0) against the committed digest.  Frames are
    # bit-identical from run to run and independent of the tile sharding (disjoint tiles, deterministic in-wave schedule), so for the
    # library build the digest was recorded with, any N must reproduce the sha256; another build may sum a pixel's samples in another
    # order, then the film's mean has to agree to 1e-5.  A mismatch fails the run: a bench line over a wrong picture is worth nothing.
    film_check = None
    if rank == 0 and frame_per_step and args.steps > 0:
        key = f"scene{args.scene} {W}x{H} {args.strategy}+{args.sampler} {spp_job}spp"
        sha, mean, mx, non_finite = film_digest(accum)
        path = os.path.join(ROOT, "profiles", "film_checksums.json")
        book = json.load(open(path)) if os.path.exists(path) else {}
        if args.write_film_checksum:
            book[key] = {"library": prod.version(), "sha256": sha, "mean": mean, "max": mx, "non_finite_values": non_finite, "n_gpus": world}
            json.dump(book, open(path, "w"), indent=1, sort_keys=True)
        ref = book.get(key)
        film_check = {"sha256": sha, "mean": round(mean, 6), "non_finite_values": non_finite, "committed": None, "bit_exact": None, "mean_rel_err": None}
        if ref:
            same_build = ref["library"] == prod.version()
            film_check.update({"committed": ref["sha256"], "committed_library": ref["library"], "bit_exact": (sha == ref["sha256"]) if same_build else None,
                               "mean_rel_err": abs(mean - ref["mean"]) / max(abs(ref["mean"]), 1e-30)})
            if (same_build and sha != ref["sha256"]) or not (film_check["mean_rel_err"] <= 1e-5) or non_finite != ref.get("non_finite_values", 0):
                print(json.dumps({"error": "film check failed", "film_check": film_check}), file=sys.stderr, flush=True)
                raise SystemExit(3)

    out = None
    if rank == 0:
        # ---- roofline: algorithmic bytes / measured kernel time (rank 0's launches) ----
        st = pkg.ffi.Stats()
        sp = pkg.make_params(spp_job, args.strategy, args.sampler, shard_index=rank, shard_count=world, collect_stats=2)
        scratch = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        prod.render_accum_device(scene, cam, sp, 0, 4, scratch.data_ptr(), stream, stats=st)
        sd = st.as_dict()
        bps = algorithmic_bytes_per_sample(sd, spp_job)
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        launch_samples = samples_per_step / world
        achieved = bps * launch_samples / (avg_ms * 1e-3) / 1e9
        algo = {"model": "production traversal (collect_stats=2): 112 B/DevNode4 visit, 48 B/triangle test, 336 B/closest hit, 16 B/spectrum, 48 B CMF",
                "bytes_per_sample": round(bps, 1), "GBps": round(achieved, 2), "frac_of_l2_request_rate": None,
                "bytes_per_launch": round(bps * launch_samples),
                "per_sample": {k: round(sd[k] / max(sd["samples"], 1), 3) for k in
                               ("closest_rays", "shadow_rays", "nodes_closest", "tris_closest", "nodes_shadow", "tris_shadow",
                                "closest_hits", "bounces", "spectrum_evals", "textured_lookups")}}
        # What bounds the kernel is VALU issue on a cache-resident working set, not HBM: `achieved` / `peak` / `frac` are
        # wave-level VALU instructions per second against the chip's issue peak.  The instruction count per sample comes from the
        # committed SQ PMC pass of THIS workload on THIS library build (tools/profile_bench.sh -> profiles/pmc_valu.json; counters
        # cannot be read from inside the process), the duration is measured live.  The SURVEY 8(d) algorithmic-bytes figure stays as
        # `algorithmic` (frac_of_hbm_peak is what round 1 reported as `frac`), the PMC HBM traffic as `traffic` / `hbm_*`.
        wl = {"scene": args.scene, "width": W, "height": H, "spp": spp_job, "spp_per_step": sps,
              "strategy": args.strategy, "sampler": args.sampler, "n_gpus": world}
        roofline = {"bound": "valu", "achieved": None, "peak": round(PEAK_VALU_GINSTR, 1), "unit": "Gwave-instr/s", "frac": None, "traffic": None,
                    "kernel": "pt_kernel<STATS=false,FEAT=scene feature mask,MODE>", "kernel_ms_avg": round(avg_ms, 3),
                    "valu": None, "algorithmic": algo}
        vj, vsrc = find_profile("pmc_valu", wl)
        if vj:
            try:
                # the count belongs to a workload; a build other than the profiled one is used too, but flagged (the count moves by a few
                # per cent between kernel versions, the duration is always live)
                ips = vj["SQ_INSTS_VALU_per_launch"] / vj["samples_per_launch"]
                ach = ips * launch_samples / (avg_ms * 1e-3) / 1e9
                lane_use = vj["SQ_THREAD_CYCLES_VALU_per_launch"] / (64.0 * vj["SQ_INSTS_VALU_per_launch"])
                roofline.update({"achieved": round(ach, 1), "frac": round(ach / PEAK_VALU_GINSTR, 4)})
                roofline["valu"] = {"wave_instr_per_sample": round(ips, 1), "issue_frac": round(ach / PEAK_VALU_GINSTR, 4),
                                    "lane_use": round(lane_use, 4), "useful_lane_frac": round(ach / PEAK_VALU_GINSTR * lane_use, 4),
                                    "source": vsrc, "profiled_library": vj.get("library"),
                                    "profile_matches_build": vj.get("library") == prod.version()}
                wc = vj.get("SQ_WAVE_CYCLES_per_launch")
                if wc:
                    roofline["valu"]["wave_cycle_shares"] = {k[3:].lower(): round(vj[k + "_per_launch"] / wc, 4) for k in
                                                             ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS")
                                                             if k + "_per_launch" in vj}
                # the memory-pipeline side of the bound (same committed passes): the vector-memory path of a CU takes one L1 tag lookup per
                # clock; a per-lane dwordx4 gather costs up to 64 of them
                if "TCP_TOTAL_CACHE_ACCESSES_sum_per_launch" in vj and vj.get("unprofiled_launch_ms"):
                    n_cu, clk = 256.0, vj.get("gpu_clock_ghz", 2.4)
                    cyc = vj["unprofiled_launch_ms"] * 1e-3 * clk * 1e9
                    l1 = {"tag_lookups_per_clk_per_cu": round(vj["TCP_TOTAL_CACHE_ACCESSES_sum_per_launch"] / n_cu / cyc, 4), "peak": 1.0,
                          "clock_ghz_assumed": clk}
                    if "TA_TA_BUSY_sum_per_launch" in vj:
                        l1["ta_busy"] = round(vj["TA_TA_BUSY_sum_per_launch"] / n_cu / cyc, 4)
                    if "SQ_INSTS_VMEM_RD_per_launch" in vj:
                        l1["lookups_per_wave_load"] = round(vj["TCP_TOTAL_CACHE_ACCESSES_sum_per_launch"] / vj["SQ_INSTS_VMEM_RD_per_launch"], 2)
                        l1["wave_loads_per_sample"] = round(vj["SQ_INSTS_VMEM_RD_per_launch"] / vj["samples_per_launch"], 2)
                    if "TCP_TCC_READ_REQ_sum_per_launch" in vj:
                        l1["l1_hit_rate"] = round(1.0 - vj["TCP_TCC_READ_REQ_sum_per_launch"] / vj["TCP_TOTAL_CACHE_ACCESSES_sum_per_launch"], 4)
                    roofline["l1"] = l1
            except Exception as e:
                roofline["valu_error"] = repr(e)
        pj, psrc = find_profile("pmc_traffic", wl)
        if pj and pj.get("library") == prod.version():
            try:
                # HBM bytes per launch from the committed rocprofv3 PMC passes (tools/profile_bench.sh: separate
                # FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 correction applied); only valid for
                # the workload and library build it was collected on
                roofline["traffic"] = pj.get("hbm_bytes_per_launch")
                roofline["traffic_source"] = psrc
                roofline["hbm_GBps"] = round(roofline["traffic"] / (avg_ms * 1e-3) / 1e9, 1)
                roofline["hbm_frac"] = round(roofline["hbm_GBps"] / PEAK_HBM_GBPS, 5)
                if "TCC_HIT_sum_per_launch" in pj:
                    req = pj["TCC_HIT_sum_per_launch"] + pj["TCC_MISS_sum_per_launch"]
                    roofline["l2_request_GBps"] = round(req * 128.0 / (avg_ms * 1e-3) / 1e9, 1)
                    roofline["l2_hit_rate"] = round(pj["TCC_HIT_sum_per_launch"] / req, 4)
                    algo["frac_of_l2_request_rate"] = round(achieved / roofline["l2_request_GBps"], 4)
            except Exception:
                pass
        # ---- CPU baseline: the oracle in faithful mode on this box's host cores (rank 0, N=1 only) ----
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import ptoracle
            orc = ptoracle.Oracle(native=True)
            osc = orc.new_scene()
            ocam = pkg.scenes.load_scene(osc, args.scene, W, H)
            orc.set_faithful(osc, True)
            cores = os.cpu_count() or 1
            import numpy as np
            tiles_x, tiles_y = (W + 7) // 8, (H + 7) // 8

            def run(k, n_idx):
                """every k-th 8x8 tile of the frame, sample indices [0, n_idx) of the same Sobol job"""
                p2 = pkg.make_params(spp_job, args.strategy, args.sampler, shard_index=0, shard_count=k)
                acc = np.zeros((H, W, 3), np.float32)
                _, sec = orc.render_accum(osc, ocam, p2, 0, n_idx, threads=cores, accum=acc)
                px = 0
                for t in range(0, tiles_x * tiles_y, k):
                    tx, ty = t % tiles_x, t // tiles_x
                    px += min(8, W - tx * 8) * min(8, H - ty * 8)
                return px * n_idx, sec

            n, sec = run(1, 1)                                    # calibration: the whole frame, one sample index
            rate = n / max(sec, 1e-9)
            want = rate * args.cpu_seconds                        # samples that fit the time budget
            if want >= 2 * W * H:                                 # whole frame, several sample indices
                k, n_idx = 1, int(min(max(want // (W * H), 1), spp_job))
            else:                                                 # keep the calibration run as the sample
                k, n_idx = 1, 1
            n, sec = run(k, n_idx)
            cpu = {"value": round(n / sec / 1e6, 5), "unit": "Msamples/s", "cores": cores, "kind": "port",
                   "sample": f"oracle faithful mode (two-level recursive BVH, non-shrinking t_max, Mat4 inverse per instance per "
                             f"ray, light sampler rebuilt per call), every {k}-th 8x8 tile of the same {W}x{H} frame, sample "
                             f"indices [0,{n_idx}) of the {spp_job}-spp Sobol job: {n} samples in {sec:.1f} s on {cores} threads"}
        out = {
            "metric": f"Msamples/sec (whole node), scene{args.scene} {args.strategy.upper()}/{args.sampler.capitalize()} {W}x{H}",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"scene{args.scene} {W}x{H} {args.strategy}+{args.sampler}, {spp_job}-spp job, "
                                   f"{sps} sample indices per step" + (" (BASELINE configs[1])" if (args.scene, W, H, spp_job) == (3, 1920, 1080, 1024) else ""),
                       "samples_per_step": samples_per_step, "seconds_to_target_spp": round(W * H * spp_job / (value * 1e6), 3),
                       "parallelism": (f"tiles8x8-rr{world}+rccl-film-reduce" if backend == "nccl" else f"tiles8x8-rr{world}+{backend}-rehearsal") if world > 1 else "single-gpu",
                       "bvh": scene_info(prod, scene)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "film_check": film_check,
        }
        if cpu:
            out["config"]["x_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def scene_info(prod, scene):
    return "flat single-level SAH BVH2 (64 B nodes with both child boxes, <=3 tris/leaf) collapsed to the 4-wide tree the kernels walk (128 B nodes); " + prod.scene_info(scene)


if __name__ == "__main__":
    main()
