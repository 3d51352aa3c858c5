#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the path tracer on the BASELINE.json workload.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--tracer megakernel|wavefront]

A "step" is one pass of the hot path over one batch of input: --spp-per-step (default 16) consecutive
samples per pixel over the whole frame, i.e. 16 of the reference's vkCmdDispatch calls
(engine/src/vk_engine.cpp:1435) handed to crt_render together, which is the unit the wavefront tracer
traces at once; the 256 spp of the headline config are 16 steps.  The per-frame-rebuild workload keeps
one frame (animate + rebuild + 1 spp) per step.  The default workload is the one
BASELINE.json's metric is quoted on, lost_empire 4K: lost_empire.obj is absent from the reference
checkout, so the documented stand-in `voxel_city(seed=1)` (224,998 triangles, one atlas texture) is
rendered at 3840x2160 and labelled as such.  With N > 1 (launched by torch.distributed.run, one rank per
GPU) the frame's 8-row bands are dealt round-robin over the ranks ("strong" scaling: the frame is fixed),
every rank holds the whole scene + BVH, and the finished image is gathered with one RCCL all-gather after
the timed steps (its time is reported as config.gather_ms, outside `value`).

rays = hit_world calls (primary + every bounce segment), counted on the device.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)



def launcher_plan(argv, n_gpus, port, base_env=None):
    """What `python bench.py --gpus N` (N > 1, not already under a launcher) starts: one child per GPU, the same script and
    arguments, with the torch.distributed.run environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns a list of
    (argv, env) pairs; kept separate from the spawning so that a CPU test can check it for N = 8."""
    env0 = dict(os.environ if base_env is None else base_env)
    plan = []
    for r in range(n_gpus):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
        plan.append(([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env))
    return plan


def requested_gpus(argv):
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    return n


PHASES = ("start", "dist_init", "scene_upload", "bvh_build", "instrumented_step", "warmup", "timed_steps", "reduce",
          "comm_init", "first_gather", "timed_gather", "fallback_gather", "gather_check", "report", "done")


class Phase:
    """Where this rank is, for the day a multi-GPU run hangs: kept in memory for the rank's own watchdog and, when the
    self-launcher started the rank, in a one-line file the launcher reads (CRT_BENCH_PHASE_DIR/rank_<r>)."""
    def __init__(self, rank):
        self.rank, self.name, self.t0 = rank, "start", time.time()
        self.pending = None      # rank 0: the finished result line while the (untimed) image gather is still running
        self.measured_t = None   # every rank: when the timed measurement (and its reduction over ranks) was complete
        d = os.environ.get("CRT_BENCH_PHASE_DIR", "")
        self.path = os.path.join(d, f"rank_{rank}") if d else None
        self.set("start")

    def set(self, name):
        assert name in PHASES, name
        self.name = name
        if self.path:
            try:
                with open(self.path, "w") as f:
                    f.write(name)
            except OSError:
                pass


def start_watchdog(phase, world, seconds):
    """A rank of a multi-GPU run that is not done after `seconds` (stuck in ncclCommInitRank, a collective, a kernel)
    prints ONE JSON line saying which rank and which phase, and ends the process with exit code 3: the driver's limit
    would otherwise kill the run with nothing on stdout.  Daemon thread; os._exit because the main thread may be inside
    a collective that never returns."""
    import threading

    def watch():
        # two limits: `seconds` for the whole run, and CRT_BENCH_GATHER_TIMEOUT (120 s) for what follows the measurement
        # (communicator set-up, the image gather, the final barrier): a hang there must not cost the result, nor minutes
        gather_limit = float(os.environ.get("CRT_BENCH_GATHER_TIMEOUT", "120"))
        while time.time() - phase.t0 < seconds and not (phase.measured_t is not None and time.time() - phase.measured_t > gather_limit):
            if phase.name == "done":
                return
            time.sleep(0.25)
        if phase.name == "done":
            return
        if phase.measured_t is not None:
            # The measurement is complete and rank 0 has (or had) the whole line; only the gather of the image or the
            # shutdown is stuck.  Rank 0 reports the line with the gather marked as failed; every rank ends with exit code 0,
            # so that a launcher (torch.distributed.run) does not turn a finished measurement into a failed run.
            pending = phase.pending
            if pending is not None:
                pending["config"]["gather"] = f"failed: timeout in phase {phase.name} after {round(time.time() - phase.measured_t, 1)} s"
                pending["config"]["gather_ms"], pending["config"]["gather_check"] = None, None
                print(json.dumps(pending), flush=True)
            print(f"bench.py: rank {phase.rank} stuck in phase {phase.name} after the measurement; ending", file=sys.stderr, flush=True)
            os._exit(0)
        print(json.dumps({"error": "timeout", "rank": phase.rank, "n_gpus": world, "phase": phase.name,
                          "after_s": round(time.time() - phase.t0, 1)}), flush=True)
        os._exit(3)

    t = threading.Thread(target=watch, daemon=True)
    t.start()
    return t


def install_term_handler(phase, world):
    """A launcher that loses one rank ends the others with SIGTERM (torch.distributed.run does).  Rank 0 then says on stdout
    what it knows: the finished line if only the image gather was outstanding, else ONE JSON line with "error", its phase
    and the time, so that a run that died on another rank does not end with an empty stdout."""
    import signal
    import threading

    def report_and_exit():
        if phase.rank == 0 and phase.name != "done":
            pending = phase.pending
            if pending is not None:
                pending["config"]["gather"] = f"failed: terminated in phase {phase.name}"
                pending["config"]["gather_ms"], pending["config"]["gather_check"] = None, None
                print(json.dumps(pending), flush=True)
                os._exit(0)
            print(json.dumps({"error": "terminated", "rank": 0, "n_gpus": world, "phase": phase.name,
                              "after_s": round(time.time() - phase.t0, 1)}), flush=True)
        os._exit(143)

    # A Python signal handler only runs when the main thread is back in the interpreter, and the main thread of a rank that
    # is being terminated sits in a collective or a stream wait.  The interpreter's C-level handler, however, runs at once
    # in whichever thread gets the signal and writes the signal number to the wake-up descriptor: a thread of its own
    # reads that pipe and reports.
    try:
        r, w = os.pipe()
        os.set_blocking(w, False)
        signal.signal(signal.SIGTERM, lambda signum, frame: report_and_exit())
        signal.set_wakeup_fd(w, warn_on_full_buffer=False)
    except (ValueError, OSError):      # not the main thread (imported by a test)
        return

    def wait_for_term():
        while True:
            b = os.read(r, 1)
            if b and b[0] == signal.SIGTERM:
                report_and_exit()

    threading.Thread(target=wait_for_term, daemon=True).start()


def rank_timeout_s():
    return float(os.environ.get("CRT_BENCH_RANK_TIMEOUT", "420"))      # below the launcher's limit, which is below the driver's (600 s in round 2)


def launch_ranks(argv, n_gpus):
    """Parent of a self-launched multi-GPU run.  It never touches the GPU (no torch, no libcrt.so: nothing has been
    imported yet), starts the ranks as child processes, waits, and fails if any of them did.  Rank 0 prints the JSON line;
    when a rank fails or the run times out the parent prints a JSON line with "error", the rank and the phase it was in."""
    import socket
    import subprocess
    import tempfile
    # a listening socket keeps the port reserved until just before the ranks start (bind-close-reuse would let another
    # process take it in between: ADVICE r2); SO_REUSEADDR lets rank 0's store bind it right after the close
    s = socket.socket()
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    phase_dir = tempfile.mkdtemp(prefix="crt_bench_phase_")
    plan = launcher_plan(argv, n_gpus, port, base_env=dict(os.environ, CRT_BENCH_PHASE_DIR=phase_dir))
    s.close()
    procs = [subprocess.Popen(a, env=e) for a, e in plan]
    deadline = time.time() + float(os.environ.get("CRT_BENCH_LAUNCH_TIMEOUT", "480"))
    codes = [None] * n_gpus
    timed_out = False
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = [c for c in codes if c not in (None, 0)]
        timed_out = time.time() > deadline
        if failed or timed_out:      # one rank died (or the run hangs): the others would wait in a collective forever
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill()                        # exactly the PIDs started above
                    p.wait()
                    codes[i] = "killed"
            break
        time.sleep(0.05)

    def phase_of(r):
        try:
            with open(os.path.join(phase_dir, f"rank_{r}")) as f:
                return f.read().strip() or "unknown"
        except OSError:
            return "not started"

    bad = [(i, c) for i, c in enumerate(codes) if c != 0]
    phases = {r: phase_of(r) for r in range(n_gpus)}
    import shutil
    shutil.rmtree(phase_dir, ignore_errors=True)
    if bad:
        own = [(i, c) for i, c in bad if c != "killed"]      # ranks that ended by themselves; the rest were ended by this launcher
        print(json.dumps({"error": "timeout" if timed_out and not own else "rank_failed", "n_gpus": n_gpus,
                          "failed_ranks": [{"rank": i, "exit": c, "phase": phases[i]} for i, c in (own or bad)],
                          "phases": {str(r): ph for r, ph in phases.items()}}), flush=True)
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and requested_gpus(sys.argv[1:]) > 1:
    # `python bench.py --gpus N` on its own: become the launcher before anything GPU-related is imported
    sys.exit(launch_ranks(sys.argv[1:], requested_gpus(sys.argv[1:])))

if __name__ == "__main__" and (int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("CRT_BENCH_FORCE_DIST", "") == "1"):
    # A rank of a multi-GPU run.  The contract is ONE JSON line on stdout, and native libraries write to file descriptor 1
    # on their own (Gloo's "[Gloo] Rank 3 is connected to ..." lines, RCCL's version banner and NCCL_DEBUG output): descriptor 1 goes to stderr
    # for them, and Python's sys.stdout (the JSON line, nothing else) keeps the original.
    sys.stdout.flush()
    _json_out = os.fdopen(os.dup(1), "w", buffering=1)
    os.dup2(2, 1)
    sys.stdout = _json_out

if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("CRT_BENCH_FORCE_DIST", "") == "1":
    # PyTorch-ROCm bundles its own HIP runtime (same soname as /opt/rocm's).  Loaded first, libcrt.so binds to that one
    # copy; loaded second, the process ends up with two runtimes and the second cannot open the GPU.
    import torch  # noqa: F401,E402

from computeraytracer_amd import abi  # noqa: E402

WORKLOADS = {
    # name: (scene, width, height, label)
    "lost_empire_4k": ("voxel_city", 3840, 2160, "lost_empire 4K stand-in voxel_city(seed=1), 224998 tris"),
    "monkey_1080p": ("monkey", 1920, 1080, "monkey_smooth.obj 1080p + ground sphere"),
    "cirno_1080p": ("cirno", 1920, 1080, "cirno_low.obj textured 1080p (earthmap substitutes the missing texture)"),
    "book1_400x225": ("book1", 400, 225, "RTIOW spheres 400x225"),
    "grid10m_1080p": ("grid10m", 1920, 1080, "10M-triangle instanced grid, per-frame LBVH rebuild"),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLOCK_HZ = 2.4e9
N_SIMD = 1024           # 256 CUs x 4 SIMDs
# VALU issue ceiling.  tools/ubench_valu.hip (profiles/r03_ubench_valu.txt) measured two issue classes on gfx950: fp32
# add / sub / mul / fma, v_mov and v_and / or / xor issue every ~2 cycles per SIMD once several waves are resident (the
# guide's "SIMD-32"), every other VALU instruction (v_cvt_f32_ubyteN, v_pk_fma_f32, min / max[3], compares, selects,
# shifts) every ~4.  A kernel's ceiling follows from its mix (tools/valu_mix.py -> profiles/r03_valu_mix.json); without
# that file the all-slow-class rate is used (0.2447 wave-instructions per clock and SIMD = 15.7 lanes).
VALU_SLOW_RATE = 0.2447
PMC_JSON = os.path.join(ROOT, "profiles", "r03_pmc.json")   # tools/pmc_to_json.py over the rocprofv3 --pmc passes of this command
MIX_JSON = os.path.join(ROOT, "profiles", "r03_valu_mix.json")


def valu_peak_tlaneops(kernel):
    """(peak in T lane-instructions/s, wave-instructions per clock and SIMD it corresponds to, where it comes from)."""
    try:
        m = json.load(open(MIX_JSON))
        c = m["kernels"][kernel]["issue_ceiling_instr_per_clk_simd"]
        src = "profiles/r03_valu_mix.json (static mix x measured class rates)" + ("" if m.get("code_rev") == abi.code_rev() else "; STALE: kernel sources changed since")
    except (OSError, ValueError, KeyError):
        c, src = VALU_SLOW_RATE, "all instructions in the 4-cycle class (profiles/r03_ubench_valu.txt)"
    return N_SIMD * 64.0 * c * CLOCK_HZ / 1e12, c, src


def pmc_counters(kernel_prefix):
    """What the committed PMC passes say about a kernel (a separate rocprofv3 --pmc run of this same command: counters
    cannot be read from inside the benchmark), per launch and per ray of THAT run: HBM-side bytes, VALU instructions
    issued (SQ_INSTS_VALU) and lanes that were live in them (SQ_THREAD_CYCLES_VALU), resident waves, L2 hit rate, busy
    fraction of the vector-memory address unit (TA).  {} when the file is missing.  "stale": the kernel sources have
    changed since the counters were taken (code_rev in the file vs. the sources beside this script)."""
    try:
        doc = json.load(open(PMC_JSON))
        ks = doc["kernels"]
    except (OSError, ValueError, KeyError):
        return {}
    best = None
    for name, d in ks.items():
        if kernel_prefix in name and "hbm_traffic_bytes_per_dispatch" in d and (best is None or d["dispatches"] > best["dispatches"]):
            best = d
    if best is None:
        return {}
    out = {"traffic": best["hbm_traffic_bytes_per_dispatch"], "l2_hit_rate": best.get("l2_hit_rate"),
           "stale": doc.get("code_rev") is None or doc.get("code_rev") != abi.code_rev()}
    c = best.get("counters_per_dispatch", {})
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0            # summed over the 8 XCDs
    if cyc > 0 and c.get("SQ_INSTS_VALU"):
        out["cycles"] = cyc
        out["valu_instr_per_clk_simd"] = round(c["SQ_INSTS_VALU"] / (N_SIMD * cyc), 4)
        out["valu_lanes_live_frac"] = round(c.get("SQ_THREAD_CYCLES_VALU", 0.0) / c["SQ_INSTS_VALU"] / 64.0, 3)
        out["valu_lane_instr"] = c.get("SQ_THREAD_CYCLES_VALU", 0.0)
        out["valu_instr"] = c["SQ_INSTS_VALU"]
        out["waves_per_simd"] = round(c.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / cyc / 1024.0, 2)
        if c.get("TA_BUSY_avr"):
            out["ta_busy_frac"] = round(c["TA_BUSY_avr"] / cyc, 3)
    return out


def kernel_roofline(name, mix_name, pmc, launch_ms, algorithmic_bytes):
    """The figures for one kernel: the roof it is under (VALU issue: live lane-instructions per second against the ceiling
    of its instruction mix), the rate at which it moves its algorithmic bytes (served mostly by L2 / Infinity Cache: NOT
    an HBM utilisation) and the HBM utilisation proper (PMC traffic / launch time / 8 TB/s).  PMC figures are per launch
    of the profiled run, which has the same launch size; launch_ms is measured live."""
    out = {"kernel": name, "launch_ms": round(launch_ms, 4)}
    if launch_ms <= 0:
        return out
    out["l2_served_GBps"] = round(algorithmic_bytes / (launch_ms * 1e-3) / 1e9, 1)
    if pmc.get("traffic"):
        out["hbm_GBps"] = round(pmc["traffic"] / (launch_ms * 1e-3) / 1e9, 1)
        out["hbm_frac"] = round(out["hbm_GBps"] / HBM_PEAK_GBS, 4)
    if pmc.get("valu_lane_instr"):
        peak, ceiling, src = valu_peak_tlaneops(mix_name)
        out["valu_Tlaneops"] = round(pmc["valu_lane_instr"] / (launch_ms * 1e-3) / 1e12, 2)
        out["valu_peak_Tlaneops"] = round(peak, 2)
        out["valu_peak_source"] = src
        out["valu_useful_frac"] = round(out["valu_Tlaneops"] / peak, 4)
        # issue rate of the live run against the mix's ceiling, and against the plain fp32-FMA rate of the guide (0.5 / clk)
        live_rate = pmc["valu_instr"] / (N_SIMD * launch_ms * 1e-3 * CLOCK_HZ)
        out["valu_instr_per_clk_simd"] = round(live_rate, 4)
        out["valu_issue_frac"] = round(live_rate / ceiling, 3)
    for k in ("valu_lanes_live_frac", "waves_per_simd", "l2_hit_rate", "ta_busy_frac"):
        if pmc.get(k) is not None:
            out[k] = pmc[k]
    if pmc.get("stale"):
        out["counters_stale"] = True
    return out


def parity_sample(scene, tracer, bvh_flags, device):
    """RMSE / primary-hit-ID check of the GPU path against the CPU oracle on a small sample of the bench scene (the
    metric names "RMSE vs ref"; the parity tests proper are tests/test_gpu_parity.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc   # the checker, never the product
    w, h, n = 256, 144, 4
    small = abi.builtin_scene(scene.name, w, h, os.path.join(ROOT, "tests", "golden"))
    seeds = abi.frame_seeds(0, n)
    ps = [small.params_for(i + 1, seeds[i]) for i in range(n)]
    ref_img, ref_ids, _ = orc.render(small, ps, w, h, flags=orc.TRAVERSE_FAST)
    with abi.Context(device) as c:
        c.configure(w, h, tracer=tracer)
        c.upload(small); c.build(bvh_flags)
        c.render(ps[0], 1, seeds); c.sync()
        img, ids = c.image(), c.hit_ids()
    a = img[..., :3] / np.maximum(img[..., 3:4], 1.0)
    b = ref_img[..., :3] / np.maximum(ref_img[..., 3:4], 1.0)
    out = {"rmse": float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))),
           "primary_id_mismatches": int((ids != ref_ids).sum()),
           "bit_exact": bool(np.array_equal(img.view(np.uint32), ref_img.view(np.uint32))),
           "sample": f"{small.name} {w}x{h} x {n} iterations vs the CPU oracle"}
    if small.world is not None:
        # reference order (threaded BVH, first primitive tested wins a tie, compute.comp:239,260,301,358) vs the build's
        # lowest-ID tie-break on the same primary rays: SURVEY 8(d) allows <= 1e-5 of them to differ, each by <= 4 ulp of t
        out["tie_break_vs_reference_order"] = orc.tie_break_report(small, ps, w, h)
    return out


def cpu_baseline(scene, target_s=12.0):
    """The CPU oracle (kind "port": there is no CPU tracer in the reference) on a bounded sample of the
    same scene/camera/seeds: 256x144, as many iterations as fit ~target_s, all host threads the box gives us."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc   # the measured baseline, never the product
    w, h = 256, 144
    small = abi.builtin_scene(scene.name, w, h, os.path.join(ROOT, "tests", "golden"))
    threads = min(orc.lib.orc_max_threads(), int(os.environ.get("OMP_NUM_THREADS", "16")))
    seeds = abi.frame_seeds(0, 2049)
    orc.render(small, [small.params_for(1, seeds[0])], w, h, flags=orc.TRAVERSE_FAST, threads=threads, want_ids=False)   # warm: BVH, threads
    t0 = time.perf_counter()
    _, _, c = orc.render(small, [small.params_for(i + 1, seeds[i]) for i in range(16)], w, h, flags=orc.TRAVERSE_FAST, threads=threads, want_ids=False)
    one = (time.perf_counter() - t0) / 16     # (a four-iteration probe read 1.8x high and the sample came out at 6.8 s instead of ~12)
    n = int(max(1, min(2048, target_s / max(one, 1e-4))))
    t0 = time.perf_counter()
    _, _, c = orc.render(small, [small.params_for(i + 2, seeds[i + 1]) for i in range(n)], w, h, flags=orc.TRAVERSE_FAST,
                         threads=threads, want_ids=False)
    dt = time.perf_counter() - t0
    out = {"value": round(c.rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"{small.name} {w}x{h} x {n} iterations ({c.rays} rays, {dt:.1f} s), oracle TRAVERSE_FAST, OpenMP"}
    if scene.world is not None and scene.n_prims <= 300000:
        # the reference builds its BVH on one host thread (B1, full-sweep SAH); restated in the oracle, whole scene
        out["bvh_build_ms"] = round(orc.ref_bvh_build_ms(scene), 1)
        out["bvh_build"] = "restated reference builder (BVHNode ctor + gpu_serialize), 1 thread, full scene"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="lost_empire_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--tracer", default=os.environ.get("CRT_TRACER", "wavefront"), choices=["megakernel", "wavefront"])
    ap.add_argument("--spp-per-step", type=int, default=16, help="iterations (samples per pixel) per step; static-scene workloads only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--refit", action="store_true", help="per-frame-rebuild workload only: crt_bvh_refit instead of crt_bvh_build in every frame")
    ap.add_argument("--bvh", default=os.environ.get("CRT_BVH", ""), choices=["", "lbvh", "sah", "collapse", "lbvh-wide", "sah-wide", "direct"],
                    help="default: sah-wide (static scenes), direct (CRT_BVH_WIDE_DIRECT) for the per-frame rebuild workload")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        # --gpus N is honoured by the self-launcher above (no WORLD_SIZE in the environment) or must agree with the
        # launcher that started this rank (torch.distributed.run --nproc-per-node N); never silently ignored
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; start it as `python bench.py --gpus {args.gpus}` "
                         f"or under torch.distributed.run with --nproc-per-node {args.gpus}")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    phase = Phase(rank)
    if world > 1:
        start_watchdog(phase, world, rank_timeout_s())
        install_term_handler(phase, world)
    # CRT_BENCH_REHEARSE=1: every rank on GPU 0 and gloo instead of RCCL, to rehearse the multi-rank path of this script
    # on a one-GPU box (RCCL refuses two ranks on one device); never set by the driver
    rehearse = os.environ.get("CRT_BENCH_REHEARSE", "") == "1"
    if rehearse:
        local_rank = 0
    # CRT_BENCH_FORCE_DIST=1: take the torch.distributed path even with one rank (RCCL init, all-reduce and all-gather
    # next to libcrt.so in one process can then be checked on a one-GPU box); never set by the driver
    if world > 1 or os.environ.get("CRT_BENCH_FORCE_DIST", "") == "1":
        import torch
        import torch.distributed as dist
        phase.set("dist_init")
        torch.cuda.set_device(local_rank)
        # Two backends in one group: gloo carries the control plane (barriers, the reductions of a few scalars, the 128-byte
        # communicator id: CPU tensors), RCCL carries the image (libcrt.so's own communicator, or torch's for CUDA tensors in
        # the fallback gather; torch makes its communicator at the first CUDA collective, not here).  The measurement then
        # does not depend on a communicator that the (untimed) gather may fail to make.
        dist.init_process_group("gloo" if rehearse else "cpu:gloo,cuda:nccl")

    scene_name, W, H, label = WORKLOADS[args.workload]
    asset_dir = os.path.join(ROOT, "tests", "golden")
    real = os.path.join(ROOT, "assets")
    if args.workload == "lost_empire_4k" and any(os.path.exists(os.path.join(real, "lost_empire" + e)) for e in (".mesh", ".obj")):
        # the real mesh when a checkout supplies it (the reference's own copy is absent)
        scene_name, asset_dir, label = "lost_empire", real, "lost_empire.obj 4K (mesh supplied under assets/)"
    scene = abi.builtin_scene(scene_name, W, H, asset_dir)
    tracer = abi.TRACER_WAVEFRONT if args.tracer == "wavefront" else abi.TRACER_MEGAKERNEL
    rebuild = args.workload == "grid10m_1080p"
    if not args.bvh:
        args.bvh = "direct" if rebuild else "sah-wide"

    phase.set("scene_upload")
    ctx = abi.Context(local_rank)
    # The per-frame-rebuild workload dispatches one small batch per frame, for which the context would otherwise measure
    # its three kernel forms over the first dispatches and keep the fastest (which one wins can differ from run to run
    # when two are close: ADVICE r2).  The benchmark fixes the form that measurement has always picked on this workload
    # (one lane per path, k_wf_tail), so that runs are reproducible; CRT_BENCH_TUNE=1 leaves the choice to the context.
    cfg_flags = abi.FLAG_PATH_KERNEL if args.workload == "grid10m_1080p" and os.environ.get("CRT_BENCH_TUNE", "") != "1" else 0
    ctx.configure(W, H, tracer=tracer, band_rank=rank, band_world=world, flags=cfg_flags)
    ctx.upload(scene)
    bvh_flags = {"lbvh": abi.BVH_LBVH, "sah": abi.BVH_SAH_REFINE, "collapse": abi.BVH_LEAF_COLLAPSE,
                 "lbvh-wide": abi.BVH_LBVH | abi.BVH_WIDE, "sah-wide": abi.BVH_SAH_REFINE | abi.BVH_WIDE, "direct": abi.BVH_WIDE_DIRECT}[args.bvh]
    wide = args.bvh.endswith("-wide") and tracer == abi.TRACER_WAVEFRONT
    phase.set("bvh_build")
    build_ms = ctx.build(bvh_flags)
    build_ms = ctx.build(bvh_flags)   # second build: allocations are warm, this is the device time of the build alone
    spp = 1 if rebuild else max(1, args.spp_per_step)
    seeds = abi.frame_seeds(0, (args.warmup + args.steps) * spp + 1)

    def run(first_step, n):
        """Steps first_step .. first_step + n - 1 (1-based); step k covers iterations (k - 1) * spp + 1 .. k * spp."""
        first_iter = (first_step - 1) * spp + 1
        if not rebuild:
            ctx.render(scene.params, first_iter, seeds[first_iter - 1:first_iter - 1 + n * spp])
            return
        for i in range(n):            # config 5: animate + full rebuild every frame, one sample per frame; nothing here waits for the GPU
            ctx.animate(float(first_iter + i), 0.35)
            if args.refit:
                ctx.refit()           # crt_bvh_refit: the boxes of the first build's tree follow the triangles
            else:
                ctx.build(bvh_flags, wait=False)
            ctx.render(scene.params, 1, seeds[first_iter - 1 + i:first_iter + i])

    # V, T, D for the bytes-per-ray model: one instrumented dispatch (untimed)
    phase.set("instrumented_step")
    ctx.stats_reset()
    ctx.set_instrumented(True)
    run(1, 1)
    ctx.sync()
    st = ctx.stats()
    ctx.set_instrumented(False)
    V = st.node_visits / max(st.rays, 1)
    T = st.prim_tests / max(st.rays, 1)
    D = st.rays / max(st.paths, 1)

    phase.set("warmup")
    run(1, args.warmup)
    ctx.sync()
    ctx.stats_reset()
    # HIP events around every k_wf_extend launch, on the stream it is launched on.  Not for the per-frame rebuild workload:
    # per-kernel timing makes every dispatch wait for the previous one and keeps small batches on the per-bounce kernels
    # (crt_set_kernel_timing, abi.h), and that workload's figure is the frame time
    # CRT_BENCH_NO_KERNEL_TIMING=1 (experiments only): whole-step time without the per-kernel events, which also lets the
    # context's small-batch kernel choice (CRT_WF_TAIL_*) apply
    kernel_timing = not rebuild and os.environ.get("CRT_BENCH_NO_KERNEL_TIMING", "") != "1"
    ctx.set_kernel_timing(kernel_timing)

    def host_barrier():
        import torch
        dist.all_reduce(torch.zeros(1, dtype=torch.float64))      # a CPU tensor: gloo (dist.barrier() may pick the CUDA backend)

    def barrier():
        if dist is not None:
            host_barrier()
            import torch
            torch.cuda.synchronize()
        ctx.sync()

    barrier()
    phase.set("timed_steps")
    t0 = time.perf_counter()
    run(args.warmup + 1, args.steps)
    ctx.sync()
    own_elapsed = time.perf_counter() - t0      # this rank's own steps, before it waits for the others: band imbalance shows here
    barrier()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()
    if rebuild and not args.refit:
        build_ms = st.last_build_ms       # the last frame's enqueue-only rebuild (device time, read after the sync)
    rays = float(st.rays)
    kernel_ms = st.last_render_ms     # HIP events on the context's stream around the timed launches (last run() call)

    gather_ms, gather_path, gather_check = None, None, None
    if dist is not None:
        import torch
        phase.set("reduce")
        tdev = torch.device("cpu")      # control plane: gloo
        t = torch.tensor([elapsed, rays], dtype=torch.float64, device=tdev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, rays = float(tmax[0]), float(t[1])
        # every rank's own time for its steps (before the closing barrier) and its ray count, gathered so the line can show
        # the spread: rows of sky and rows of city cost differently, and the slowest rank sets `value`
        per = torch.zeros(2 * world, dtype=torch.float64, device=tdev)
        per[2 * rank], per[2 * rank + 1] = own_elapsed, float(st.rays)
        dist.all_reduce(per, op=dist.ReduceOp.SUM)
        per = per.cpu().numpy().reshape(world, 2)
        rank_ms = [float(x) / args.steps * 1e3 for x in per[:, 0]]
        per_rank_rays = [float(x) for x in per[:, 1]]
    def gather_phase():
        import torch
        tdev = torch.device("cpu")      # control plane: gloo
        gather_ms, gather_check = None, None
        # gather the finished image: ONE all-gather of the packed band buffers + un-interleave, outside `value`
        gather_path = "rccl-native"
        if rehearse:
            # gloo rehearsal on one GPU (RCCL refuses two ranks per device): the torch plumbing of dist.py
            gather_path = "torch-gloo-rehearsal"
            from computeraytracer_amd import dist as cdist
            band = cdist.context_band_tensor(ctx, W, torch.device("cuda", local_rank)).to(tdev)
            torch.cuda.synchronize(); host_barrier()
            g0 = time.perf_counter()
            image = cdist.gather_image(band, H, world)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - g0) * 1e3
            assert tuple(image.shape) == (H, W, 4)
            full = image.cpu().numpy()
        else:
            # the product path: crt_comm_init + crt_image_gather inside libcrt.so (csrc/dist/gather_rccl.hip); torch.distributed
            # only carries the 128-byte communicator id from rank 0 to the others.  The gather is outside `value`, so a
            # communicator that cannot be made must not cost the run its result: every step that can raise is followed by an
            # agreement (all-reduce MIN of "ok") before anybody enters the next collective, and if the native path is not
            # available on every rank the image is gathered through torch.distributed instead and the line says so.
            def agree(ok):
                f = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=tdev)
                dist.all_reduce(f, op=dist.ReduceOp.MIN)
                return float(f[0]) == 1.0
            native_error = None
            phase.set("comm_init")
            idt = torch.zeros(abi.COMM_ID_BYTES, dtype=torch.uint8, device=tdev)
            ok = True
            if rank == 0:
                try:
                    idt.copy_(torch.frombuffer(bytearray(abi.comm_unique_id()), dtype=torch.uint8))
                except Exception as e:      # noqa: BLE001 -- reported in the line
                    ok, native_error = False, f"crt_comm_unique_id: {e}"
            ok = agree(ok)
            if ok:
                dist.broadcast(idt, 0)
                try:
                    ctx.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
                except Exception as e:      # noqa: BLE001
                    ok, native_error = False, f"crt_comm_init: {e}"
                ok = agree(ok)
            if ok:
                phase.set("first_gather")
                ctx.image_gather(); ctx.sync()            # the first collective on a communicator sets its channels up: untimed
                torch.cuda.synchronize(); host_barrier()
                phase.set("timed_gather")
                g0 = time.perf_counter()
                ctx.image_gather(); ctx.sync()
                gather_ms = (time.perf_counter() - g0) * 1e3
                full = ctx.image_gathered()
                ctx.comm_destroy()
            else:
                phase.set("fallback_gather")
                gather_path = "torch-nccl-fallback (native communicator unavailable" + (": " + native_error if native_error else " on another rank") + ")"
                from computeraytracer_amd import dist as cdist
                band = cdist.context_band_tensor(ctx, W, torch.device("cuda", local_rank))
                torch.cuda.synchronize(); host_barrier()
                g0 = time.perf_counter()
                image = cdist.gather_image(band, H, world)
                torch.cuda.synchronize()
                gather_ms = (time.perf_counter() - g0) * 1e3
                full = image.cpu().numpy()
        # every rank's bands arrived: each pixel of the rendered grid carries the last iteration's number in .w, and
        # this rank's own rows are the ones it rendered
        phase.set("gather_check")
        last_iter = float((args.warmup + args.steps) * spp) if not rebuild else 1.0
        own = abi.band_rows(H, rank, world)
        gh, gw = (H // 8) * 8, (W // 8) * 8
        ok = bool((full[:gh, :gw, 3] == last_iter).all()) and bool(np.array_equal(full[own], ctx.image()[own]))
        okt = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=tdev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        gather_check = "ok" if float(okt[0]) == 1.0 else "MISMATCH"
        gt = torch.tensor([gather_ms], dtype=torch.float64, device=tdev)
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        gather_ms = float(gt[0])

        return gather_ms, gather_path, gather_check

    phase.set("report")
    if rank == 0:
        mrays = rays / elapsed / 1e6
        # bytes/ray of the whole pipeline = 32 V + 48 T + 96 (48-byte queue entry written + read) + 32 / D (accumulation)
        # node bytes per box tested: 32 (pair BVH: 32-byte nodes) or 8 (8-wide quantised nodes: 64 bytes loaded per 8 boxes)
        node_b = 8.0 if wide else 32.0
        bytes_per_ray = node_b * V + 48.0 * T + (96.0 if tracer == abi.TRACER_WAVEFRONT else 0.0) + 32.0 / D
        rank_rays = float(st.rays)
        if tracer == abi.TRACER_WAVEFRONT and st.last_extend_launches:
            # dominant kernel k_wf_extend: of that figure it moves 32 V + 48 T + 32 (ray read) + 8 (hit record written)
            kernel_name, kernel_bytes_per_ray = "k_wf_extend", node_b * V + 48.0 * T + 40.0
            n_launches = float(st.last_extend_launches) if not rebuild else float(st.last_extend_launches) * args.steps
            launch_ms = st.last_extend_ms / st.last_extend_launches
            # bounce 0 runs in k_wf_primary (packets of primary rays) when the tree is 8-wide: its rays (one per path) and
            # its time are not k_wf_extend's
            if st.last_primary_launches:
                rank_rays -= float(st.paths)
        else:
            kernel_name = "k_megakernel" if tracer == abi.TRACER_MEGAKERNEL else "k_wf_* (one render call, kernels not timed one by one)"
            kernel_bytes_per_ray = bytes_per_ray
            n_launches = float(args.steps * spp)
            launch_ms = kernel_ms / (args.steps * spp) if not rebuild else kernel_ms
        launch_rays = rank_rays / n_launches
        achieved = launch_rays * kernel_bytes_per_ray / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        line = {
            "metric": "Mrays/s (whole node) lost_empire 4K 256spp; BVH build ms; RMSE vs ref",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "scene": scene_name, "width": W, "height": H, "spp_per_step": spp,
                       "tris": int(len(scene.tris)), "spheres": int(len(scene.spheres)), "quads": int(len(scene.quads)),
                       "kernel_form": {0: "per-bounce (batches) / measured (single samples)", abi.FLAG_PATH_KERNEL: "path kernel, one lane per path (fixed)"}[cfg_flags],
                       "tracer": args.tracer, "bvh": args.bvh + ("+refit per frame" if rebuild and args.refit else ""), "rng": "ref-hash", "parallelism": f"tile{world}" if world > 1 else "single",
                       "bvh_build_ms": round(build_ms, 4), "gather_ms": None if gather_ms is None else round(gather_ms, 3),
                       "gather": gather_path, "gather_check": gather_check,
                       "rays_per_path": round(D, 3), "nodes_per_ray": round(V, 2), "prim_tests_per_ray": round(T, 2)},
            "roofline": None,
        }
        # The dominant kernel's roof.  k_wf_extend is bound by VALU issue under wave64 divergence (and, as busy, by the
        # vector-memory address path), not by HBM: its algorithmic bytes are served by L2 / Infinity Cache.  `frac` is
        # therefore the fraction of the chip's VALU lane slots that do useful work; the algorithmic-bytes rate is kept as
        # l2_served_GBps and the HBM utilisation proper as hbm_frac.  The PMC part comes from the committed counter
        # passes of this same command (profiles/r02_pmc.json) and is attached for the workload they were taken on.
        pmc_ok = args.workload == "lost_empire_4k" and scene_name == "voxel_city" and kernel_name == "k_wf_extend" and args.bvh == "sah-wide"
        pmc = pmc_counters("k_wf_extend<false") if pmc_ok else {}
        rl = kernel_roofline(kernel_name, "k_wf_extend", pmc, launch_ms, launch_rays * kernel_bytes_per_ray)
        rl.update({"kernel_bytes_per_ray": round(kernel_bytes_per_ray, 1), "pipeline_bytes_per_ray": round(bytes_per_ray, 1),
                   "launches": int(n_launches), "rays_per_launch": round(launch_rays)})
        if tracer == abi.TRACER_WAVEFRONT and st.last_primary_launches:
            fused = st.last_shade_launches == st.last_extend_launches
            pm_ms = st.last_primary_ms / st.last_primary_launches
            pm_rays = float(st.paths) / st.last_primary_launches
            ppmc = pmc_counters("k_wf_primary<") if pmc_ok else {}
            # bounce 0 per path: 8-wide nodes and triangles as in the model, no queue entry read (the kernel makes its own camera
            # rays); shading record 64 B, 48 B written per surviving path, 16 B per ended one
            prim = kernel_roofline("k_wf_primary" + (" (rays, traversal and shading of bounce 0)" if fused else ""), "k_wf_primary", ppmc, pm_ms,
                                   pm_rays * (node_b * V + 48.0 * T + 64.0 + 48.0 * (1.0 - 1.0 / D) + 16.0 / D))
            prim.update({"launches": int(st.last_primary_launches), "rays_per_launch": round(pm_rays)})
            rl["primary"] = prim
        if rl.get("valu_useful_frac") is not None:
            line["roofline"] = dict({"bound": "valu-issue", "achieved": rl["valu_Tlaneops"], "peak": rl["valu_peak_Tlaneops"], "unit": "Tlane-op/s",
                                     "frac": rl["valu_useful_frac"], "traffic": pmc.get("traffic")}, **rl)
            line["roofline"]["counters_source"] = "profiles/r03_pmc.json (rocprofv3 --pmc passes of this command)"
            if pmc.get("stale"):
                line["roofline"]["counters_stale"] = True
        else:
            # no counter passes for this workload / tracer: only the algorithmic-bytes rate is known, and it is NOT an HBM
            # utilisation (the scene lives in L2 / Infinity Cache); reported against the HBM peak as the contract asks
            line["roofline"] = dict({"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                                     "note": "algorithmic bytes / launch time; served mostly by L2 / Infinity Cache, not an HBM utilisation"}, **rl)
        if pmc_ok and st.last_shade_launches:
            spmc = pmc_counters("k_wf_shade")
            shade_ms = st.last_shade_ms / st.last_shade_launches
            # k_wf_shade per path: 56 B queue entry + hit read, 64 B shading record, 48 B written per surviving path, 16 B per ended one
            # every bounce has a k_wf_shade, except bounce 0 when k_wf_primary runs the whole first bounce (then there are as many
            # k_wf_shade launches as k_wf_extend launches and the primary rays are not theirs)
            fused_first = st.last_primary_launches and st.last_shade_launches == st.last_extend_launches
            shade_rays = (float(st.rays) - (float(st.paths) if fused_first else 0.0)) / st.last_shade_launches
            line["roofline"]["shade"] = kernel_roofline("k_wf_shade", "k_wf_shade", spmc, shade_ms, shade_rays * (56.0 + 64.0 + 48.0 * (1.0 - 1.0 / D) + 16.0 / D))
        # BVH build against SURVEY 8(d)'s 376 B / primitive model
        n_prims = int(len(scene.tris) + len(scene.spheres) + len(scene.quads))
        line["build"] = {"ms": round(build_ms, 4), "prims": n_prims, "bytes_per_prim": 376,
                         "frac": round(376.0 * n_prims / (build_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 5) if build_ms > 0 else None}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene)
            line["parity"] = parity_sample(scene, tracer, bvh_flags, local_rank)
        if dist is not None:
            # per-rank spread: ms per step of every rank's own steps, and its share of the rays
            line["config"]["rank_ms_per_step"] = {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4),
                                                  "all": [round(x, 4) for x in rank_ms]}
            line["config"]["rank_rays_share"] = {"min": round(min(per_rank_rays) / max(sum(per_rank_rays), 1.0), 5),
                                                 "max": round(max(per_rank_rays) / max(sum(per_rank_rays), 1.0), 5)}
    if dist is not None:
        # The gather comes last and is outside `value`: rank 0's line is complete by now, and the watchdog prints it (with
        # the gather marked as failed) if a rank gets stuck in the communicator set-up or in the collective.
        if rank == 0:
            phase.pending = line
        phase.measured_t = time.time()
        try:
            gather_ms, gather_path, gather_check = gather_phase()
        except Exception as e:      # noqa: BLE001 -- the result must not be lost to the untimed gather
            if rank == 0:
                line["config"]["gather"] = f"failed in phase {phase.name}: {e}"
                line["config"]["gather_ms"], line["config"]["gather_check"] = None, None
                print(json.dumps(line), flush=True)
            print(f"bench.py: rank {rank}: image gather failed in phase {phase.name}: {e}", file=sys.stderr, flush=True)
            os._exit(0)             # other ranks may be inside a collective that will never complete; the measurement stands
        if rank == 0:
            phase.pending = None
            line["config"]["gather_ms"] = None if gather_ms is None else round(gather_ms, 3)
            line["config"]["gather"] = gather_path
            line["config"]["gather_check"] = gather_check
    if rank == 0:
        print(json.dumps(line), flush=True)
    ctx.close()
    if dist is not None:
        host_barrier()
        dist.destroy_process_group()
    phase.set("done")


if __name__ == "__main__":
    main()
