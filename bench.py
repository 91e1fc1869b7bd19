#!/usr/bin/env python3
"""Headline benchmark: grid-points/s regridded, O1280 -> 0.25 degree x 137 levels.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        # no launcher: this process starts the N workers itself (launch_workers) and relays the line

A *step* is one pass of the regrid hot path over one batch of synthetic input already resident in HBM: ONE 137-level stack on
the O1280 octahedral grid (6 599 680 points) is interpolated to the 0.25 degree lat-lon grid (1 038 240 points) with k = 4
inverse-distance weights from cKDTree (BASELINE.json configs[2], the configuration the metric is quoted on), float64 — the
reference's own arithmetic (R: fields.py:178-202; `--dtype f32` for the other width) — one `atx_regrid_ell` launch per rank.

N > 1 (STRONG scaling, one process per GPU): the job is the SAME job — that one stack, the loop the reference runs at
R: filters/fields/regrid.py:204-208 — with its target points sharded N ways: every rank interpolates its traffic-balanced 1/N slice
of the target grid in one launch per step, no collective in the data path.  `value` = the point-fields of the whole job (1 038 240 x
137 per step, whatever N) / the max-over-ranks wall time of the K steps, with the source stack resident on every rank (that is what
"inputs resident in HBM" means for a target-sharded job): value(N) / value(1) is the speed-up of a fixed amount of work and can
fall short of N — per-rank launches get short (0.1 ms at N = 8), the slowest shard sets the step.  north_star's ">= 6x at 8 GPUs"
is THIS ratio.  What it costs to GET the sources there is measured in the same run and reported next
to `value`, never inside it: `source_exchange_ms.broadcast` (N RCCL broadcasts of one stack each), `.all_gather` (the same as one all-gather), `.bands`
(band-limited RCCL send/recv), each verified bit-equal against the stacks the rank synthesised itself, and
`end_to_end` — one step of the N-stack job INCLUDING the exchange, broadcast r+1 overlapped with launch r — and `end_to_end_bands`, the same step
with the band-limited all-to-all in front of one batched launch.  `weak` is the per-GPU-work-fixed line of rounds 1-4 (N stacks per
step, every rank its 1/N slice of all N in one batched launch: approaches N by construction), `config4` is BASELINE configs[3] on
the N GPUs of the run (O1280 -> N320-sized, 24 stacks resident per rank, target points over the ranks) and `field_axis_sharding`
the exchange-free alternative.
Host-side barriers and the max-over-ranks reduction run on a gloo group, stack traffic on an nccl (= RCCL) group.

Also on the same JSON line:
  roofline     algorithmic bytes of one launch / its average HIP-event duration against the 8 TB/s HBM3E peak
  cpu_baseline the oracle's scipy `csr_array @ x` statement (the reference's CPU path, R: filters/fields/regrid.py:310)
               timed on one host core on a bounded sample of the same workload (rank 0, N = 1 only)
  extras       (N = 1) k = 1, f64, field-major, fused epilogue, BASELINE configs 2 and 4, the device k-NN build, all-core CPU
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

T_IMPORT = time.monotonic()  # before the heavy imports: the first `import torch` on a fresh box takes one to two minutes
try:  # how old the process already was when this file started to run (interpreter start, launcher hand-over)
    import psutil

    T_PROCESS_AGE = max(0.0, time.time() - psutil.Process().create_time())
except Exception:  # noqa: BLE001 - a clock offset, never a reason to fail
    try:
        T_PROCESS_AGE = max(0.0, time.time() - os.stat(f"/proc/{os.getpid()}").st_ctime)
    except OSError:
        T_PROCESS_AGE = 0.0

import numpy as np
import torch


def wall_seconds() -> float:
    """Seconds since this PROCESS started — what a driver's time limit around the command counts."""
    return T_PROCESS_AGE + (time.monotonic() - T_IMPORT)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
SEED = 20260630


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--src-grid", default="o1280")
    ap.add_argument("--tgt-grid", default="0.25")
    ap.add_argument("--levels", type=int, default=137)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--dtype", default="f64", choices=["f32", "f64"],
                    help="arithmetic of the headline: f64 is the reference's own (R: fields.py:178-202 hands float64 to numpy / scipy); "
                         "the other width is reported in extras")
    ap.add_argument("--layout", default="columns", choices=["columns", "fields"])
    ap.add_argument("--tile", type=int, default=0, help="targets per workgroup (0 = library default)")
    ap.add_argument("--natural-order", action="store_true", help="visit the targets in row-major order instead of column blocks (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary lines (N = 1: extras; N > 1: exchange / end-to-end / strong)")
    ap.add_argument("--headline-shape-only", action="store_true",
                    help="extras: keep the lines measured on the headline's own grids (k = 1, the other width, field-major, fused epilogue, k-NN "
                         "build) and skip BASELINE configs 2 / 4 / 5 and the all-core CPU child, which run at their own full sizes")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--cpu-workers", type=int, default=0,
                    help="worker processes of extras.cpu_all_cores (0 = all cores this process may use: os.cpu_count() capped by the cgroup quota)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N > 1: backend of the group that carries the stacks (nccl = RCCL over xGMI); barriers always run on gloo")
    ap.add_argument("--secondary-seconds", type=float, default=360.0,
                    help="N > 1: wall-clock budget of the lines measured after `value`; when it runs out the JSON line is printed with what is there")
    ap.add_argument("--total-seconds", type=float, default=300.0,
                    help="N > 1: cap on the wall clock of the WHOLE command, process start to the printed line: a secondary section whose "
                         "estimated cost (SECTION_ESTIMATE_S) no longer fits is `skipped`; `value` is measured before any of them")
    ap.add_argument("--rehearse-multi", action="store_true",
                    help="REHEARSAL ONLY, with --gpus 1: run the N > 1 secondary sections (RCCL exchange through torch.distributed and through "
                         "atx_comm_*, end to end, strong) at world size 1 — the whole multi-GPU code path on the real collective library")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher check (runs without a GPU): every rank joins the host-side gloo group, one barrier and one max-reduction, "
                         "rank 0 prints {\"launch_check\": true, ...} — what `python bench.py --gpus N` does before it touches the device")
    ap.add_argument("--share-device", action="store_true",
                    help="REHEARSAL ONLY: all ranks use cuda:0 (exercises the N > 1 code path on a 1-GPU box; needs --backend gloo)")
    return ap.parse_args()


def synth_stack(grid, n_lev, dtype, dev, stack_id, layout):
    """v[l, p] = 280 + 30 sin(lat) cos(2 lon + 0.1 l) + N(0, 1)   (SURVEY.md §8d), built in HBM; a pure function of stack_id."""
    from anemoi_transform_amd.stack import COLUMNS, Stack

    n_pts = len(grid["latitudes"])
    lat = torch.from_numpy(np.deg2rad(grid["latitudes"])).to(dev)
    lon = torch.from_numpy(np.deg2rad(grid["longitudes"])).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED + 1000 * stack_id)
    st = Stack.empty(n_pts, n_lev, dtype, dev, layout, zero=True)
    sin_lat = torch.sin(lat)
    for l in range(n_lev):
        v = 280.0 + 30.0 * sin_lat * torch.cos(2.0 * lon + 0.1 * l + 0.37 * stack_id)
        v = v + torch.randn(n_pts, dtype=torch.float64, device=dev, generator=gen)
        if layout == COLUMNS:
            st.data[:, l] = v.to(dtype)
        else:
            st.data[l, :n_pts] = v.to(dtype)
    return st


def algorithmic_bytes(n_lev, itemsize, n_unique, n_tgt, k):
    """L*B*(U + Nt) + Nt*k*4 + [k>1]*Nt*k*B."""
    b = n_lev * itemsize * (n_unique + n_tgt) + n_tgt * k * 4
    if k > 1:
        b += n_tgt * k * itemsize
    return b


def ordered_tables(idx, w, grid, lo, hi, np_dtype, dev, natural=False):
    """Device tables of the targets [lo, hi) in the order the library's policy visits them (gather.target_order_for: column blocks
    of the target grid for rows of 5 or more neighbours, natural order for the headline's k = 4) + the output row of every table
    row; (idx, w, None) in natural order."""
    from anemoi_transform_amd.gather import target_order_for

    order = None if natural else target_order_for(grid["latitudes"][lo:hi], grid["longitudes"][lo:hi], idx.shape[1] if idx.ndim == 2 else 1)
    pick = slice(None) if order is None else order
    idx_d = torch.from_numpy(np.ascontiguousarray(idx[lo:hi][pick]).astype(np.int32)).to(dev)
    w_d = torch.from_numpy(np.ascontiguousarray(w[lo:hi][pick]).astype(np_dtype)).to(dev)
    return idx_d, w_d, (None if order is None else torch.from_numpy(order).to(dev))


def launch_times(fn, steps, warmup):
    """HIP-event durations (ms) of `steps` calls of `fn` (one launch each) on the current stream — the stream libatx launches on."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


def time_launches(fn, steps, warmup):
    """Average and minimum of `launch_times`."""
    ms = launch_times(fn, steps, warmup)
    return float(np.mean(ms)), float(np.min(ms))


class quiet_stdout:
    """File descriptor 1 points at stderr inside the block: the collective libraries print connection notes to stdout
    ("[Gloo] Rank 0 is connected to ..."), and stdout is reserved for the ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def host_description() -> dict:
    """SURVEY.md §8(d) "CPU baseline beside it": what the CPU figures were measured ON — CPU model (/proc/cpuinfo, as lscpu prints
    it), logical cores of the host, the cores this process may actually use (affinity, cgroup quota), numpy / scipy / python."""
    import platform

    import scipy

    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for row in f:
                if row.lower().startswith("model name"):
                    model = row.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    quota = None
    try:  # cgroup v2 CPU quota of the box, in cores
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(period)
    except Exception:  # noqa: BLE001 - a description, never a reason to fail
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return {"model": model or platform.processor() or "unknown", "nproc": os.cpu_count(), "usable_cores": usable, "cpu_quota_cores": quota,
            "numpy": np.__version__, "scipy": scipy.__version__, "python": platform.python_version(), "machine": platform.machine()}


def mirror_multi_gpu_into_config(result: dict) -> None:
    """The driver's record keeps `config` and `roofline` whole and reduces every other extra key to its name: the numbers that say
    what the N > 1 job costs INCLUDING its source exchange, and the fixed-total-work line, are mirrored into `config.multi_gpu`."""
    def pick(section, *keys):
        rec = result.get(section)
        if not isinstance(rec, dict):
            return None
        if "error" in rec or "skipped" in rec:
            return {k: rec[k] for k in ("error", "skipped") if k in rec}
        return {k: rec.get(k) for k in keys}

    exchange = result.get("source_exchange_ms") or {}
    result["config"]["multi_gpu"] = {
        "strong": pick("strong", "value", "ms_per_step"),
        "weak": pick("weak", "value", "ms_per_step", "stacks_per_step"),
        "end_to_end": pick("end_to_end", "value", "ms_per_step", "verified_bit_equal"),
        "end_to_end_bands": pick("end_to_end_bands", "value", "ms_per_step", "verified_bit_equal"),
        "end_to_end_all_gather": pick("end_to_end_all_gather", "value", "ms_per_step", "verified_bit_equal"),
        "source_exchange_ms": {"broadcast": exchange.get("broadcast"), "all_gather": exchange.get("all_gather"), "bands": exchange.get("bands")},
        "field_axis_sharding": pick("field_axis_sharding", "value", "ms_per_step"),
        "config4": pick("config4", "value", "ms_per_step"),
        "config5": pick("config5", "value", "ms_per_step"),
        "secondary_timed_out_in": result.get("secondary_timed_out_in"),
        "secondary_error": result.get("secondary_error"),
        # the wall clock of the command (seconds, max over ranks): where the time of an N-rank run goes, and what was dropped to stay under the cap
        "wall_s": result.get("wall_s"),
        "setup_s": result.get("setup_s"),
        "setup_s_rank": result.get("setup_s_rank"),
        "precompute_s": result.get("precompute_s"),
        "sections_s": result.get("sections_s"),
        "sections_skipped": result.get("sections_skipped"),
        "total_seconds_cap": result.get("total_seconds_cap"),
        "unit": "grid-points/s (values), ms (times), s (wall_s / setup_s / precompute_s / sections_s)",
    }


def carried_traffic(key: str) -> dict | None:
    """The PMC record of one launch shape from profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_probe.py,
    calibrated on atx_stream_copy) — measured under the profiler in its own run, CARRIED here, never measured by this process."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[key]
    except Exception:  # noqa: BLE001 - absent file or key: nothing to carry
        return None
    return {"hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": rec["algorithmic_bytes_per_launch"],
            "traffic_over_algorithmic": rec["traffic_over_algorithmic"], "measured": rec.get("measured"),
            "source": f"carried from profiles/traffic.json[{key!r}] (tools/pmc_probe.py under rocprofv3 --pmc), not re-measured in this run"}


def line(n_units, ms, alg_bytes):
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {"value": n_units / (ms * 1e-3), "avg_launch_ms": ms, "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS}


def launch_workers(args) -> int:
    """`python bench.py --gpus N` without a launcher: this process — which has not touched the GPU and never will — starts N
    FRESH worker processes through torch.distributed.run (one per GPU, rendezvous on 127.0.0.1 at a free port), passes rank 0's
    single JSON line through on its own stdout and returns the launcher's exit status.  No process that has initialised HIP is
    ever replaced by another program."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this platform; must be set before a worker's HIP runtime starts
    env.setdefault("GLOO_SOCKET_IFNAME", "lo")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)  # stderr passes straight through
    lines = [l for l in child.stdout.splitlines() if l.strip().startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    if child.returncode != 0:
        print(f"bench.py: the {args.gpus}-rank job ended with status {child.returncode}", file=sys.stderr)
        return child.returncode
    return 0 if lines else 4


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:  # the driver's plain `python bench.py --gpus N`
        raise SystemExit(launch_workers(args))
    # before anything initialises the HIP runtime (it reads its environment once)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node by contract: the host-side group talks over loopback, whatever the hostname resolves to
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if args.launch_check:
        import torch.distributed as dist

        with quiet_stdout():
            dist.init_process_group("gloo")
            dist.barrier()
            t = torch.tensor([float(rank)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "max_rank_seen": int(t.item()),
                              "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}), flush=True)
        with quiet_stdout():
            dist.barrier()
            dist.destroy_process_group()
        return

    setup_s = {"process_start_to_main": wall_seconds()}  # interpreter start + `import torch` (one to two minutes on a fresh box)
    marks = {"t": time.monotonic()}

    def mark(name):  # seconds since the previous mark, under `name`
        now = time.monotonic()
        setup_s[name] = setup_s.get(name, 0.0) + (now - marks["t"])
        marks["t"] = now

    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.interp import knn_inverse_distance
    from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack

    native.load()  # fails loudly if the HIP extension is missing
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", 0 if args.share_device else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    multi = world > 1 or args.rehearse_multi
    if multi:
        import torch.distributed as dist

        assert not (args.share_device and args.backend == "nccl"), "RCCL needs one GPU per rank"
        if world == 1:  # --rehearse-multi without a launcher
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        with quiet_stdout():
            dist.init_process_group("gloo")  # host-side: barriers, the max-over-ranks of the elapsed time
            dist.barrier()

    def barrier():
        if multi:
            dist.barrier()

    def max_over_ranks(seconds: float) -> float:
        if not multi:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    tdtype = torch.float32 if args.dtype == "f32" else torch.float64
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    itemsize = 4 if args.dtype == "f32" else 8
    layout = COLUMNS if args.layout == "columns" else FIELDS
    if args.tile:
        native.set_tuning(args.tile)
    mark("library_device_and_host_group")

    # ---- one-off precompute on the host (not timed): grids, cKDTree indices + weights
    t0 = time.perf_counter()
    src_grid, tgt_grid = lookup(args.src_grid), lookup(args.tgt_grid)
    n_src, n_tgt = len(src_grid["latitudes"]), len(tgt_grid["latitudes"])
    idx64, w64 = knn_inverse_distance(src_grid, tgt_grid, k=args.k)
    precompute_s = time.perf_counter() - t0
    n_unique = int(np.unique(idx64).size)
    mark("grids_and_knn_table")

    # target-point shard of this rank: contiguous, balanced by HBM traffic (GatherPlan.bounds) — equal-count
    # shards of a lat-lon target are 1.8x apart in cost (polar targets share their source columns)
    plan = GatherPlan(n_src, n_tgt, index=idx64, weights=w64)
    if layout == COLUMNS and not args.natural_order:  # shards of the plan (the N > 1 sections) keep the visiting order of their targets
        from anemoi_transform_amd.gather import target_order_for

        plan.order_targets(target_order_for(tgt_grid["latitudes"], tgt_grid["longitudes"], args.k))
    # the headline step is ONE short launch per rank (0.1 ms at 8 ranks): cut with the weight measured for that shape
    # (gather.TARGET_COST_SHORT_LAUNCH, profiles/r05_shard_weight_sweep.log); the weak / exchange sections keep the library's default cut
    from anemoi_transform_amd.gather import TARGET_COST_SHORT_LAUNCH

    bounds = plan.bounds(world, target_cost=TARGET_COST_SHORT_LAUNCH)
    lo, hi = bounds[rank], bounds[rank + 1]
    mark("plan_and_shard_bounds")

    # ---- sources resident in HBM before the timed region: the N stacks of the step, each a pure function of its id
    stacks = [synth_stack(src_grid, args.levels, tdtype, dev, r, layout) for r in range(world)]
    # the tables in the order the library's policy visits the targets (natural for the headline's k = 4; column blocks of the target
    # grid from k = 5 on: GatherPlan.order_targets / atx_regrid_ell_ordered, same results bit for bit)
    idx_d, w_d, rows_d = ordered_tables(idx64, w64, tgt_grid, lo, hi, np_dtype, dev, natural=layout != COLUMNS or args.natural_order)
    assert native.check_indices(idx_d, n_src) == 0
    outs = [Stack.empty(hi - lo, args.levels, tdtype, dev, layout)]
    torch.cuda.synchronize()
    mark("source_stacks_and_tables_resident")

    def launch(src, out, idx=None, w=w_d, k=args.k, n_t=hi - lo):
        rows = rows_d if idx is None else None  # callers that bring their own tables bring them in natural order
        native.regrid_ell(src.data, out.data, idx_d if idx is None else idx, w if k > 1 else None, n_src=n_src, n_tgt=n_t, k=k,
                          n_lev=src.n_lev, src_pitch=src.pitch, out_pitch=out.pitch, layout=src.layout, tgt_rows=rows)

    def step():  # the job of EVERY N: stack 0 -> this rank's slice of the target grid, one launch (fixed total work: strong scaling)
        launch(stacks[0], outs[0])

    # ---- timed region: W warm-up steps, then exactly K steps between barriers
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = max_over_ranks(elapsed)

    units_per_step = n_tgt * args.levels  # all ranks together: ONE stack x the full target grid (the shards tile it), whatever N
    value = units_per_step * args.steps / elapsed
    mark("timed_region")
    if rank == 0:  # the headline's ingredients, on stderr the moment they exist: a run cut short later has still said them
        print(f"bench.py: value {value:.6g} grid-points/s, {elapsed / args.steps * 1e3:.4f} ms/step over {args.steps} steps on {world} GPU(s), "
              f"{wall_seconds():.1f} s after process start", file=sys.stderr, flush=True)

    # ---- roofline of the dominant kernel, HIP events around single launches (N > 1: this rank's shard of the stack)
    launch_ms = launch_times(step, min(max(args.steps, 10), 200), 2)
    avg_ms, min_ms, median_ms = float(np.mean(launch_ms)), float(np.min(launch_ms)), float(np.median(launch_ms))
    shard_unique = int(np.unique(idx64[lo:hi]).size)
    alg = algorithmic_bytes(args.levels, itemsize, shard_unique, hi - lo, args.k)
    achieved = alg / (avg_ms * 1e-3) / 1e9
    # the same figure for the WHOLE job: the algorithmic bytes of all ranks' launches of one step over the step time the driver's value is
    # computed from (max over ranks, wall clock of the K steps) against N x the peak — "fraction of the HBM roofline at 1, 2, 4, 8 GPUs"
    alg_all = float(alg)
    if multi:
        t_alg = torch.tensor([alg_all], dtype=torch.float64)
        dist.all_reduce(t_alg, op=dist.ReduceOp.SUM)
        alg_all = float(t_alg.item())
    job_gbs = alg_all / (elapsed / args.steps) / 1e9
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath))
            key = f"{args.src_grid}->{args.tgt_grid} k={args.k} L={args.levels} {args.dtype} {args.layout} gpus={world}"
            if key in rec:
                traffic = rec[key].get("hbm_bytes_per_launch")
                # NOT measured by this process: PMC counters need rocprofv3 around the run (tools/pmc_probe.py)
                traffic_source = ("carried from profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same "
                                  f"launch ({rec[key].get('measured', rec.get('_measured', 'round 1'))}), not re-measured in this run")
        except Exception:
            traffic = None

    result = {
        "metric": "grid-points/sec regridded, O1280->0.25deg x137 levels",
        "value": value,
        "unit": "grid-points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        # a property of the SERIES the driver builds from the N = 1, 2, 4, 8 lines: the total work of a step is the same on every
        # line (one stack, the full target grid), so value(N) / value(1) is a speed-up that can fail to reach N
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"{args.src_grid.upper()} ({n_src} pts) -> {args.tgt_grid} deg lat-lon ({n_tgt} pts), "
                        f"k={args.k} inverse-distance regrid x {args.levels} levels per stack",
            "layout": args.layout,
            "target_order": "natural (row-major)" if rows_d is None else "column blocks of the target grid (results identical; atx_regrid_ell_ordered: the library's policy for k >= 5)",
            "stacks_per_step": 1,
            "sharding": ("target points over ranks (contiguous, traffic-balanced with the weight measured for one short launch per rank), no collective in the data path; FIXED TOTAL WORK: every line of the "
                         "N = 1, 2, 4, 8 series regrids the same ONE stack (BASELINE configs[2], the loop at R: regrid.py:204-208) per step, each rank "
                         "its 1/N slice of the target points, the source stack resident on every rank before the timed region (inputs resident in "
                         "HBM) — `value` is STRONG scaling and EXCLUDES the source exchange: value(N) / value(1) is the driver's scaling figure, it is "
                         "what answers north_star's '>= 6x at 8 GPUs', and it can fall short (short per-rank launches, the slowest shard sets the "
                         "step).  config.multi_gpu beside it: `strong` repeats `value`; `weak` is rounds 1-4's line (N stacks per step, per-GPU work "
                         "fixed: approaches N by construction); `end_to_end` (one step of the N-stack job INCLUDING the RCCL broadcast of the N "
                         "stacks, overlapped with the launches), `end_to_end_all_gather` (one all-gather instead) and `end_to_end_bands` (the "
                         "band-limited all-to-all) are the rates of a job that must move its sources every step — 'the source broadcast once via "
                         "RCCL' of north_star priced per step; `source_exchange_ms` is the once-only cost a resident job pays up front")
                        if multi else "single GPU: the N = 1 point of the fixed-total-work series (one stack per step on every line)",
            "launches_per_step_per_gpu": 1,
            **({"collectives": f"stacks: {args.backend} group ({'RCCL over xGMI' if args.backend == 'nccl' else 'gloo'}); "
                               "barriers and the max-over-ranks of the elapsed time: gloo group"} if multi else {}),
            **({"rehearsal": "ranks share one GPU over gloo; not a scaling measurement"} if args.share_device else {}),
            **({"rehearsal": "N > 1 sections at world size 1 (RCCL code path on one GPU); not a scaling measurement"} if args.rehearse_multi else {}),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "regrid_cols_ell_direct_kernel" if layout == COLUMNS else "regrid_fields_ell_kernel",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": alg,
            "avg_launch_ms": avg_ms,  # `achieved` is priced on the AVERAGE; median and minimum beside it ("median and min reported")
            "median_launch_ms": median_ms,
            "min_launch_ms": min_ms,
            "distinct_source_points": shard_unique,
            # all ranks together, on the wall clock of the timed region (what `value` is computed from); at N = 1 the launch-gap-inclusive twin of `frac`
            "job": {"algorithmic_bytes_per_step_all_ranks": alg_all, "achieved": job_gbs, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                    "frac": job_gbs / (HBM_PEAK_GBS * world), "ms_per_step": elapsed / args.steps * 1e3},
        },
        "precompute_s": max_over_ranks(precompute_s),  # host-side k-NN table (cKDTree or its cache file), slowest rank
    }
    mark("roofline_launches")
    if multi:
        # where the wall clock of an N-rank command goes before the secondary sections start: the stage times of the rank that took
        # LONGEST to get here, as that rank saw them (an entry-by-entry maximum would count a slow rank's stage and the others' wait
        # for it at the next barrier twice)
        keys = sorted(setup_s)
        mine = torch.tensor([setup_s[k] for k in keys], dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        slowest = max(range(world), key=lambda r: float(every[r].sum()))
        setup_s = {k: round(float(v), 3) for k, v in zip(keys, every[slowest].tolist())}
        result["setup_s_rank"] = slowest
    result["setup_s"] = setup_s
    result["total_seconds_cap"] = args.total_seconds if multi else None

    if multi:
        # the fixed-total-work line IS the headline: repeated under the name the rounds 1-4 records used for it (before the secondary
        # sections, so that a line cut short by their budget carries it too)
        result["strong"] = {"value": value, "unit": "grid-points/s", "ms_per_step": elapsed / args.steps * 1e3, "scaling": "strong",
                            "note": "the headline itself: ONE 137-level stack (the N = 1 job), each rank its 1/N target slice; source resident on every rank"}
    if multi and not args.no_extras:
        # RCCL announces itself on stdout when its first communicator comes up (version, host, library path): the secondary sections
        # run with file descriptor 1 pointing at stderr, stdout stays reserved for the ONE JSON line
        quiet = quiet_stdout()
        quiet.__enter__()
        try:
            multi_gpu_lines(args, result, dist, rank, world, dev, plan, stacks, launch, units_per_step * world, barrier, max_over_ranks,
                            src_grid, tgt_grid, n_src, n_tgt, tdtype, np_dtype, idx64, w64, layout, quiet.saved)
        except Exception as e:  # noqa: BLE001
            # An error OUTSIDE a section (sections catch their own): the other ranks are, or will be, inside a collective this rank will
            # never join — no further collective from here.  The measured line goes out (rank 0) and the process ends with a failure
            # status; the others leave through their watchdog the same way.
            import traceback

            traceback.print_exc(file=sys.stderr)
            if rank == 0:
                result["secondary_error"] = f"{type(e).__name__}: {e}"
                result["wall_s"] = wall_seconds()
                mirror_multi_gpu_into_config(result)
                os.write(quiet.saved, (json.dumps(result) + "\n").encode())
            os._exit(3)
        finally:
            quiet.__exit__()

    if rank == 0 and world == 1 and not args.rehearse_multi:
        single_gpu_lines(args, result, dev, stacks, outs, launch, idx_d, w_d, rows_d, idx64, w64, n_src, n_tgt, n_unique, alg, src_grid, tgt_grid,
                         tdtype, np_dtype, itemsize, plan)

    result["wall_s"] = max_over_ranks(wall_seconds())  # process start -> this line, slowest rank
    if multi:
        mirror_multi_gpu_into_config(result)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if multi:
        barrier()
        with quiet_stdout():
            dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------------
# N > 1: what is measured AFTER `value` — the exchange over RCCL, end to end, strong scaling, field-axis sharding
# ------------------------------------------------------------------------------------------------------------------------
# What a secondary section is EXPECTED to cost on the slowest rank of an N-rank run at the default (full) sizes, in seconds: the budget
# rule of `section` skips a section whose estimate no longer fits under --total-seconds.  (base, per rank): the base is what the section
# took at world 1 on an MI355X box on real RCCL (profiles/r06_bench_rehearse_multi_world1.json, `sections_s`: weak 0.18, broadcast 0.92,
# end_to_end 0.03, bands 0.02, config4 0.90, config5 1.91, field axis 0.19, c_abi init 0.05; the whole command 9.3 s) times FIVE and
# rounded up — eight ranks share one host's cores, page cache and PCIe while they build their tables — and the per-rank part covers what
# grows with N: N stacks to move per exchange (7.2 GB each), N communicators to bring up.  An unknown name costs 5 s.  The estimates only
# matter within their own length of the cap: a run that is nowhere near 300 s skips nothing.
SECTION_ESTIMATE_S = {
    "weak": (1.0, 0.25),
    "exchange broadcast": (5.0, 2.0),   # first collective on the data group: RCCL brings up its rings here
    "end_to_end": (1.0, 0.5),
    "exchange bands": (1.0, 0.5),
    "end_to_end_bands": (1.0, 0.5),
    "config4": (10.0, 0.5),             # cKDTree for the N320-sized target + 88.7 GB of sources made resident + the launches
    "config5": (10.0, 0.5),             # device k-NN over O2560 + a 29 GB stack + the launches
    "exchange all_gather": (1.0, 1.0),
    "end_to_end_all_gather": (1.0, 1.0),
    "field_axis_sharding": (1.0, 0.25),
    "c_abi init": (2.0, 1.0),
    "c_abi broadcast": (1.0, 0.5),
    "c_abi all_gather": (1.0, 1.0),
    "c_abi bands": (1.0, 0.5),
    "c_abi end_to_end": (1.0, 0.5),
}


def section_estimate(name: str, world: int) -> float:
    base, per_rank = SECTION_ESTIMATE_S.get(name, (5.0, 0.0))
    return base + per_rank * world


def multi_gpu_lines(args, result, dist, rank, world, dev, plan, stacks, launch, units_per_step, barrier, max_over_ranks,
                    src_grid, tgt_grid, n_src, n_tgt, tdtype, np_dtype, idx64, w64, layout, real_stdout_fd):
    from anemoi_transform_amd import distributed as atxd
    from anemoi_transform_amd import native
    from anemoi_transform_amd.stack import COLUMNS, Stack

    # A hung collective must not cost the line that is already measured: when the budget runs out every rank leaves, and
    # rank 0 prints what it has.  (Exceptions are caught per section; this is for waits that never return.)
    state = {"section": "start"}
    if os.environ.get("ATX_BENCH_TEST_FAULT") == "outside_sections":  # test hook (tests/test_gpu_bench_contract.py)
        raise RuntimeError("injected: a failure outside every section")

    def give_up():
        try:
            if rank == 0:
                payload = None
                for _ in range(50):  # the main thread may be adding a key to `result` this very moment: serialise a consistent view
                    try:
                        result["secondary_timed_out_in"] = state["section"]
                        result["wall_s"] = wall_seconds()  # (this rank's clock: nobody waits for the others any more)
                        mirror_multi_gpu_into_config(result)
                        payload = json.dumps(result)
                        break
                    except RuntimeError:  # "dictionary changed size during iteration"
                        time.sleep(0.01)
                if payload is not None:
                    os.write(real_stdout_fd, (payload + "\n").encode())  # fd 1 points at stderr in here
        finally:
            os._exit(3)  # the measured line is out, but a collective that never returned is a FAILURE of the run: the launcher must see it

    # the hard limit (a collective that never returns): the secondary budget, and no later than half a cap past the cap — but never less
    # than a minute from here: a run that is ALREADY past its cap skips every section, which still takes a few barrier round trips
    watchdog = threading.Timer(max(0.0, min(args.secondary_seconds, max(60.0, 1.5 * args.total_seconds - wall_seconds()))), give_up)
    watchdog.daemon = True
    watchdog.start()

    t_secondary = time.monotonic()
    sections_s: dict[str, float] = {}
    skipped: list[str] = []
    result["sections_s"] = sections_s  # seconds per secondary section, slowest rank — filled as they complete
    result["sections_skipped"] = skipped

    def section(name, fn):
        """Run one secondary measurement on all ranks; an exception on any rank is recorded and the ranks stay in step.
        A section is SKIPPED (all ranks decide together, on the slowest rank's clock) when its estimated cost no longer fits under
        `--total-seconds` counted from process start, or past half of the secondary budget: the hard limit — which ends the run with a
        non-zero status — is for a collective that never returns, not for a slow box."""
        state["section"] = name
        barrier()
        clock = torch.tensor([time.monotonic() - t_secondary, wall_seconds()], dtype=torch.float64)
        dist.all_reduce(clock, op=dist.ReduceOp.MAX)
        spent, wall = clock.tolist()
        estimate = section_estimate(name, world)
        why = None
        if wall + estimate > args.total_seconds:
            why = (f"{wall:.0f} s after process start; this section is estimated at {estimate:.0f} s and the command is capped at "
                   f"{args.total_seconds:.0f} s (--total-seconds)")
        elif spent > 0.5 * args.secondary_seconds:
            why = f"{spent:.0f} s of the {args.secondary_seconds:.0f} s secondary budget already spent"
        if why is not None:
            skipped.append(name)
            return {"skipped": why}
        t_section = time.monotonic()
        try:
            out = fn()
            ok = 1.0
        except Exception as e:  # a comparison line must never take the bench line down
            out, ok = {"error": f"{type(e).__name__}: {e}"}, 0.0
        torch.cuda.synchronize()
        done = torch.tensor([-ok, time.monotonic() - t_section], dtype=torch.float64)
        dist.all_reduce(done, op=dist.ReduceOp.MAX)  # (max of -ok = -min of ok)
        sections_s[name] = round(float(done[1].item()), 3)
        if -done[0].item() < 1.0 and ok == 1.0:
            out = {"error": "failed on another rank"}
        return out

    # this rank's slice under the library's DEFAULT cut (GatherPlan.bounds(world): what distributed.py's functions take from the plan),
    # its tables and one output stack per source stack — the weak step below fills them, the exchange sections verify against them
    lo_w, hi_w = plan.shard_range(rank, world)
    idx_w, w_w, rows_w = ordered_tables(idx64, w64, tgt_grid, lo_w, hi_w, np_dtype, dev, natural=layout != COLUMNS or args.natural_order)
    outs = [Stack.empty(hi_w - lo_w, args.levels, tdtype, dev, layout) for _ in stacks]

    def weak_step():  # rounds 1-4's step: one launch over the N stacks (atx_regrid_ell_batch / _ordered: grid.y = stack)
        if len(stacks) == 1 or layout != COLUMNS:
            for s_, o_ in zip(stacks, outs):
                native.regrid_ell(s_.data, o_.data, idx_w, w_w, n_src=n_src, n_tgt=hi_w - lo_w, k=args.k, n_lev=args.levels, src_pitch=s_.pitch,
                                  out_pitch=o_.pitch, layout=layout, tgt_rows=rows_w)
        else:
            native.regrid_ell_batch([s.data for s in stacks], [o.data for o in outs], idx_w, w_w, n_src=n_src, n_tgt=hi_w - lo_w,
                                    k=args.k, n_lev=args.levels, src_pitch=stacks[0].pitch, out_pitch=outs[0].pitch, layout=layout, tgt_rows=rows_w)

    # ---- weak scaling (rounds 1-4's `value`): N stacks per step, every rank its 1/N target slice of ALL of them in one batched launch.
    #      Per-GPU work is that of the N = 1 job, so this approaches N x by construction; it also leaves in `outs` this rank's slice of
    #      every stack, which the exchange sections below verify their results against.  (`units_per_step` in here: the N-stack job's.)
    def weak():
        for _ in range(args.warmup):
            weak_step()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            weak_step()
        torch.cuda.synchronize()
        t = max_over_ranks(time.perf_counter() - t0)
        return {"value": units_per_step * args.steps / t, "unit": "grid-points/s", "ms_per_step": t / args.steps * 1e3, "scaling": "weak",
                "stacks_per_step": world,
                "note": "N stacks per step (N variables x 137 levels), each rank its 1/N target slice of all N in ONE batched launch; sources resident on every rank"}

    result["weak"] = section("weak", weak)

    # ---- field-axis sharding (axis 2): every rank regrids its own stack to the WHOLE target grid — no exchange at all
    def field_axis():
        full_idx = torch.from_numpy(idx64.astype(np.int32)).to(dev)
        full_w = torch.from_numpy(w64.astype(np_dtype)).to(dev)
        own, full_out = stacks[rank], Stack.empty(n_tgt, args.levels, tdtype, dev, layout)

        def own_step():
            launch(own, full_out, idx=full_idx, w=full_w, n_t=n_tgt)

        for _ in range(args.warmup):
            own_step()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            own_step()
        torch.cuda.synchronize()
        t = max_over_ranks(time.perf_counter() - t0)
        return {"value": units_per_step * args.steps / t, "unit": "grid-points/s", "ms_per_step": t / args.steps * 1e3,
                "note": "each rank regrids its own stack to the full target grid; no source exchange"}

    if layout != COLUMNS:  # field-major stacks: the exchange helpers of distributed.py are written for column stacks
        result["field_axis_sharding"] = section("field_axis_sharding", field_axis)
        watchdog.cancel()
        return

    # ---- BASELINE configs[3] as specified: O1280 -> N320-sized, 6 variables x 137 levels x 4 timesteps = 24 stacks, the target points
    #      sharded over the N ranks (no exchange in the step: every rank holds the 24 source stacks, 88.7 GB f32)
    def config4():
        from anemoi_transform_amd.gather import GatherPlan
        from anemoi_transform_amd.grids import lookup
        from anemoi_transform_amd.interp import knn_inverse_distance

        g_src, g_tgt = lookup("o1280"), lookup("n320-sized")
        # one stack per timestep, the 6 variables of a grid point sharing a column (6 x 137 = 822 levels, 3.3 KB): the layout a job
        # that controls its stacks would choose — a gathered column wastes less of its first and last 128-byte line
        # (extras.config4 of the N = 1 line times both this and 24 stacks of 137 levels)
        n4_src, n4_tgt, n_var, n_stack = len(g_src["latitudes"]), len(g_tgt["latitudes"]), 6, 4
        lev4 = n_var * args.levels
        idx4, w4 = knn_inverse_distance(g_src, g_tgt, k=4)
        b4 = GatherPlan(n4_src, n4_tgt, index=idx4, weights=w4).bounds(world)
        lo4, hi4 = b4[rank], b4[rank + 1]
        gen = torch.Generator(device=dev)
        srcs = []
        t4, np4 = torch.float32, np.float32  # SURVEY.md §8d sizes config 4 in float32 (86.8 GB of sources per GPU; float64 would not leave room beside the N headline stacks)
        for i in range(n_stack):
            gen.manual_seed(SEED + 7 * i)
            st = Stack.empty(n4_src, lev4, t4, dev, COLUMNS, zero=True)
            st.data[:, :lev4].normal_(250.0 + 5.0 * i, 20.0, generator=gen)
            srcs.append(st)
        dsts = [Stack.empty(hi4 - lo4, lev4, t4, dev, COLUMNS) for _ in range(n_stack)]
        idx_d4, w_d4, rows_d4 = ordered_tables(idx4, w4, g_tgt, lo4, hi4, np4, dev, natural=args.natural_order)

        def go():
            native.regrid_ell_batch([s.data for s in srcs], [d.data for d in dsts], idx_d4, w_d4, n_src=n4_src, n_tgt=hi4 - lo4, k=4,
                                    n_lev=lev4, src_pitch=srcs[0].pitch, out_pitch=dsts[0].pitch, layout=COLUMNS, tgt_rows=rows_d4)

        reps = max(3, min(args.steps, 20))
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            go()
        torch.cuda.synchronize()
        t = max_over_ranks(time.perf_counter() - t0)
        return {"value": n4_tgt * lev4 * n_stack * reps / t, "unit": "grid-points/s", "ms_per_step": t / reps * 1e3, "fields": n_stack * lev4,
                "targets_of_this_rank": hi4 - lo4,
                "dtype": "f32",
                "workload": f"O1280 -> N320-sized ({n4_tgt} pts), k=4, float32, {n_var} variables x {args.levels} levels x {n_stack} timesteps resident on every rank "
                            f"as {n_stack} stacks of {lev4} levels, target points over {world} ranks"}

    def config5():
        plain, fused, _, all_units, _, keep = config5_case(args, dev, tdtype, np_dtype, rank, world)
        reps = max(3, min(args.steps, 20))
        for _ in range(3):
            fused()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fused()
        torch.cuda.synchronize()
        t = max_over_ranks(time.perf_counter() - t0)
        return {"value": all_units * reps / t, "unit": "grid-points/s", "ms_per_step": t / reps * 1e3, "scaling": "strong",
                "workload": f"O2560 -> 0.25 deg, k=4, 137 levels, regrid | orog_to_z | convert fused in one launch; one stack, target points over {world} ranks"}

    # ---- the source exchange itself, on the data group (RCCL): first use creates the communicators
    state["section"] = "data group"
    t_group = time.monotonic()
    group_error = None
    try:
        if os.environ.get("ATX_BENCH_TEST_FAULT") == "data_group":  # test hook (tests/test_gpu_bench_contract.py)
            raise RuntimeError("injected: the collective library did not come up")
        atxd.set_data_group(dist.new_group(backend=args.backend))
    except Exception as e:  # noqa: BLE001 - the collective library failing to come up must not cost the measured line
        group_error = f"{type(e).__name__}: {e}"
    ok = torch.tensor([0.0 if group_error else 1.0], dtype=torch.float64)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # (on the host-side gloo group: all ranks take the same way from here)
    sections_s["data group"] = round(max_over_ranks(time.monotonic() - t_group), 3)
    if ok.item() < 1.0:
        result["source_exchange"] = {"error": group_error or "the data group failed on another rank"}
        # what needs no exchange still runs
        result["config4"] = section("config4", config4)
        torch.cuda.empty_cache()
        result["config5"] = section("config5", config5)
        torch.cuda.empty_cache()
        result["field_axis_sharding"] = section("field_axis_sharding", field_axis)
        watchdog.cancel()
        return
    if "error" in result["weak"] or "skipped" in result["weak"]:  # the exchange sections verify against `outs`: fill them whatever became of the timing
        weak_step()
        torch.cuda.synchronize()
    mine = stacks[rank]
    exchange_ms, verified, detail = {}, {}, {}
    result["source_exchange_ms"] = exchange_ms  # filled as the sections complete: a budget cut keeps what is there
    result["source_exchange"] = detail

    def timed(fn):
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return out, max_over_ranks(time.perf_counter() - t0) * 1e3

    def broadcast():
        atxd.warm_up_transport()  # communicator set-up (rings, point-to-point channels) with one-element messages
        got, ms = timed(lambda: atxd.exchange_stacks(mine))
        exchange_ms["broadcast"] = ms
        # every received stack must be the one this rank synthesised for that id
        verified["broadcast"] = all(torch.equal(g.data, s.data) for g, s in zip(got, stacks))
        return {"ms": ms, "bytes_received_per_gpu": (world - 1) * mine.data.numel() * mine.data.element_size(),
                "verified_bit_equal": verified["broadcast"]}

    def bands():
        (got, local_plan), ms = timed(lambda: atxd.exchange_source_bands(mine, plan))
        exchange_ms["bands"] = ms
        # the banded plan on the received slabs must reproduce this rank's slice of every stack, bit for bit (outs: the timed region's)
        banded = local_plan.apply_many(got)
        verified["bands"] = all(torch.equal(b.data, o.data) for b, o in zip(banded, outs))
        return {"ms": ms, "bytes_received_per_gpu": sum(b.data.numel() * b.data.element_size() for i, b in enumerate(got) if i != rank),
                "verified_bit_equal": verified["bands"]}

    def all_gather():
        got, ms = timed(lambda: atxd.exchange_stacks(mine, collective="all_gather"))
        exchange_ms["all_gather"] = ms
        return {"ms": ms, "bytes_received_per_gpu": (world - 1) * mine.data.numel() * mine.data.element_size(),
                "verified_bit_equal": all(torch.equal(g.data, s.data) for g, s in zip(got, stacks)),
                "note": "the whole-stack exchange as ONE all-gather instead of N broadcasts"}

    # ---- ORDER = PRIORITY.  The sections run in the order below and the cap drops them from the END: first what north_star words ("the
    #      source field broadcast once via RCCL", then the step that includes it), then the xGMI-native band exchange, then BASELINE
    #      configs[3] and [4] on the N GPUs, then the all-gather and field-axis comparison points, last the same exchanges once more
    #      through the C-ABI communicator.
    detail["broadcast"] = section("exchange broadcast", broadcast)

    def repetitions(exchange):  # as many repetitions as fit ~3 s of exchange; none if one exchange alone takes more than 30 s (gloo rehearsals)
        ms = exchange_ms.get(exchange)
        if ms is None or ms > 30000.0:
            return 0
        return max(1, min(3, int(3000.0 / max(ms, 1.0))))

    def end_to_end():
        reps = repetitions("broadcast")
        if reps == 0:
            return {"skipped": "the broadcast exchange failed or took more than 30 s"}
        (got, ms) = timed(lambda: [atxd.pipelined_sharded_regrid(plan, mine) for _ in range(reps)][-1])
        same = all(torch.equal(g.data, o.data) for g, o in zip(got, outs))
        return {"ms_per_step": ms / reps, "value": units_per_step / (ms / reps * 1e-3), "unit": "grid-points/s", "verified_bit_equal": same,
                "note": "one step INCLUDING the exchange of the N source stacks: broadcast r+1 (RCCL) overlapped with launch r, two source buffers alive"}

    result["end_to_end"] = section("end_to_end", end_to_end)
    detail["bands"] = section("exchange bands", bands)

    def end_to_end_bands():
        # the xGMI-native form of the same step: ONE all-to-all of band slabs (every rank sends each peer only the source columns that
        # peer's target slice references — 7 point-to-point transfers per GPU, all links at once), then the batched launch on the slabs
        reps = repetitions("bands")
        if reps == 0:
            return {"skipped": "the band-limited exchange failed or took more than 30 s"}

        def once():
            got, local_plan = atxd.exchange_source_bands(mine, plan)
            return local_plan.apply_many(got)

        (got, ms) = timed(lambda: [once() for _ in range(reps)][-1])
        same = all(torch.equal(g.data, o.data) for g, o in zip(got, outs))
        return {"ms_per_step": ms / reps, "value": units_per_step / (ms / reps * 1e-3), "unit": "grid-points/s", "verified_bit_equal": same,
                "note": "one step INCLUDING the exchange, band-limited: all-to-all of the source slabs each target slice needs, then one batched launch"}

    result["end_to_end_bands"] = section("end_to_end_bands", end_to_end_bands)

    def end_to_end_all_gather():
        # the whole-stack exchange as ONE all-gather, then one batched launch over this rank's target slice of the N stacks
        reps = repetitions("all_gather")
        if reps == 0:
            return {"skipped": "the all-gather exchange failed or took more than 30 s"}
        local = plan.shard(rank, world)

        def once():
            return local.apply_many(atxd.exchange_stacks(mine, collective="all_gather"))

        (got, ms) = timed(lambda: [once() for _ in range(reps)][-1])
        same = all(torch.equal(g.data, o.data) for g, o in zip(got, outs))
        return {"ms_per_step": ms / reps, "value": units_per_step / (ms / reps * 1e-3), "unit": "grid-points/s", "verified_bit_equal": same,
                "note": "one step INCLUDING the exchange: one all-gather of the N source stacks, then one batched launch"}

    torch.cuda.empty_cache()
    result["config4"] = section("config4", config4)
    torch.cuda.empty_cache()
    result["config5"] = section("config5", config5)
    torch.cuda.empty_cache()
    detail["all_gather"] = section("exchange all_gather", all_gather)
    torch.cuda.empty_cache()
    result["end_to_end_all_gather"] = section("end_to_end_all_gather", end_to_end_all_gather)
    torch.cuda.empty_cache()
    result["field_axis_sharding"] = section("field_axis_sharding", field_axis)

    # ---- the same exchanges through the library's own C-ABI communicator (atx_comm_*: RCCL bound directly)
    if args.backend == "nccl":
        c_abi = {}
        detail["c_abi"] = c_abi
        holder = {}

        def c_abi_init():
            holder["comm"] = atxd.atx_comm_from_torch()  # the 128-byte id travels over the gloo group; ncclCommInitRank on every rank
            return {"rccl_version": native.Comm.rccl_version()}

        def c_abi_broadcast():
            comm = holder["comm"]
            got, ms = timed(lambda: atxd.exchange_stacks(mine, comm=comm))
            return {"ms": ms, "verified_bit_equal": all(torch.equal(g.data, s.data) for g, s in zip(got, stacks))}

        def c_abi_all_gather():
            comm = holder["comm"]
            got, ms = timed(lambda: atxd.exchange_stacks(mine, comm=comm, collective="all_gather"))
            return {"ms": ms, "verified_bit_equal": all(torch.equal(g.data, s.data) for g, s in zip(got, stacks))}

        def c_abi_bands():
            comm = holder["comm"]
            (got, local_plan), ms = timed(lambda: atxd.exchange_source_bands(mine, plan, comm=comm))
            banded = local_plan.apply_many(got)
            return {"ms": ms, "verified_bit_equal": all(torch.equal(b.data, o.data) for b, o in zip(banded, outs))}

        def c_abi_end_to_end():
            comm = holder["comm"]
            reps = max(1, repetitions("broadcast"))
            (got, ms) = timed(lambda: [atxd.pipelined_sharded_regrid(plan, mine, comm=comm) for _ in range(reps)][-1])
            return {"ms_per_step": ms / reps, "value": units_per_step / (ms / reps * 1e-3), "unit": "grid-points/s",
                    "verified_bit_equal": all(torch.equal(g.data, o.data) for g, o in zip(got, outs))}

        c_abi["init"] = section("c_abi init", c_abi_init)
        if "error" not in c_abi["init"] and "skipped" not in c_abi["init"]:  # (the same on every rank: `section` agrees on both)
            c_abi["broadcast"] = section("c_abi broadcast", c_abi_broadcast)
            torch.cuda.empty_cache()
            c_abi["all_gather"] = section("c_abi all_gather", c_abi_all_gather)
            torch.cuda.empty_cache()
            c_abi["bands"] = section("c_abi bands", c_abi_bands)
            c_abi["end_to_end"] = section("c_abi end_to_end", c_abi_end_to_end)
            try:
                holder["comm"].destroy()
            except Exception as e:
                c_abi["destroy"] = {"error": f"{type(e).__name__}: {e}"}
    watchdog.cancel()


# ------------------------------------------------------------------------------------------------------------------------
# N = 1: parity spot check, CPU baseline, secondary kernel lines, BASELINE configs 2 and 4
# ------------------------------------------------------------------------------------------------------------------------
def single_gpu_lines(args, result, dev, stacks, outs, launch, idx_d, w_d, rows_d, idx64, w64, n_src, n_tgt, n_unique, alg, src_grid, tgt_grid,
                     tdtype, np_dtype, itemsize, plan):
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.gather import TARGET_COST_SHORT_LAUNCH, GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack

    from oracle import oracle  # checker / baseline only

    n_sample = min(8, args.levels)
    levels = np.linspace(0, args.levels - 1, n_sample).astype(int)
    sample64 = np.stack([stacks[0].level_numpy(int(l)).astype(np.float64) for l in levels])
    got = np.stack([outs[0].level_numpy(int(l)) for l in levels])
    indptr = (np.arange(n_tgt + 1, dtype=np.int64) * args.k).astype(np.int32)
    indices = idx64.astype(np.int32).reshape(-1)
    want = np.stack([
        oracle.csr_apply(w64.astype(np_dtype).reshape(-1), indices, indptr, (n_tgt, n_src), f.astype(np_dtype))
        for f in sample64
    ])
    result["parity_max_rel_err"] = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-30)))

    if not args.no_cpu_baseline:
        torch.set_num_threads(1)
        # SURVEY.md §8(d)(A) "as the reference runs": one process, one thread, a Python loop over fields executing the reference's
        # two regrid statements — `csr_array @ x` (R: regrid.py:310) at the bench's k and `x[..., idx]` (R: regrid.py:380, k = 1) —
        # in float64 (the reference's dtype, to_numpy default) and in float32.  `value` is the statement the headline replaces.
        from scipy.sparse import csr_array

        idx1_host = idx64[:, 0].astype(np.int64)  # cKDTree hands int64 indices to the fancy index
        budget = max(args.cpu_seconds, 0.1)
        share = {"csr_f64": 0.5, "csr_f32": 0.2, "k1_f64": 0.15, "k1_f32": 0.15}
        variants = {}

        def sample(name, statement, fields, what):
            statement(fields[0])
            n_done, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget * share[name]:
                for f in fields:
                    _ = statement(f)
                    n_done += 1
            s = time.perf_counter() - t0
            variants[name] = {"value": n_done * n_tgt / s, "unit": "grid-points/s", "ms_per_field": s / n_done * 1e3, "fields": n_done,
                              "seconds": s, "statement": what}

        matrix64 = csr_array((w64.reshape(-1), indices, indptr), shape=(n_tgt, n_src))
        matrix32 = csr_array((w64.astype(np.float32).reshape(-1), indices, indptr), shape=(n_tgt, n_src))
        sample32 = sample64.astype(np.float32)
        sample("csr_f64", lambda f: matrix64 @ f, sample64, f"scipy csr_array(k={args.k}) @ x, float64 (R: regrid.py:310)")
        sample("csr_f32", lambda f: matrix32 @ f, sample32, f"scipy csr_array(k={args.k}) @ x, float32 matrix and field")
        sample("k1_f64", lambda f: f[..., idx1_host], sample64, "x[..., nearest_grid_points], float64 (R: regrid.py:380)")
        sample("k1_f32", lambda f: f[..., idx1_host], sample32, "x[..., nearest_grid_points], float32")
        head = variants["csr_f64"]
        result["cpu_baseline"] = {
            "value": head["value"],
            "unit": "grid-points/s",
            "cores": 1,
            "kind": "port",
            "dtype": "f64",
            "sample": f"{head['fields']} fields ({n_sample} distinct levels of the same synthetic stack, float64 as in the "
                      f"reference) x scipy csr_array(k={args.k}) @ x, {head['seconds']:.1f} s on 1 thread; the k = 1 statement and the "
                      f"float32 forms of both in `variants` ({sum(v['seconds'] for v in variants.values()):.1f} s in all); "
                      f"host has {os.cpu_count()} logical cores",
            "ms_per_field": head["ms_per_field"],
            "host": host_description(),
            "variants": variants,
        }

    if args.no_extras:
        return
    extras = {}
    n_lev = args.levels
    # k = 1 nearest-neighbour gather (R: regrid.py:380), same stack
    idx1 = torch.from_numpy(idx64[:, 0].astype(np.int32).copy()).to(dev)  # (natural order: the k = 1 lines are the plain gather)
    ms1, _ = time_launches(lambda: launch(stacks[0], outs[0], idx=idx1, w=None, k=1, n_t=n_tgt), 10, 2)
    extras["nearest_k1"] = line(n_tgt * n_lev, ms1, algorithmic_bytes(n_lev, itemsize, int(np.unique(idx64[:, 0]).size), n_tgt, 1))
    # config-5 shape on the same stack: regrid -> orog_to_z -> convert fused in ONE launch
    prog = native.level_program([[(native.OP_MUL, 0, 9.80665, 0.0)] * n_lev, [(native.OP_AFFINE, 0, 1.0, -273.15)] * n_lev], dev)
    msf, _ = time_launches(lambda: native.regrid_ell(
        stacks[0].data, outs[0].data, idx_d, w_d, n_src=n_src, n_tgt=n_tgt, k=args.k, n_lev=n_lev,
        src_pitch=stacks[0].pitch, out_pitch=outs[0].pitch, layout=stacks[0].layout, prog=prog, n_stage=2, tgt_rows=rows_d), 10, 2)
    extras["fused_regrid_orog_to_z_convert"] = line(n_tgt * n_lev, msf, alg)

    # ---- the N = 2, 4, 8 lines of the strong-scaling series, rank by rank on THIS GPU: every rank's traffic-balanced shard of the
    #      headline launch timed alone (source resident, as in the N-rank run).  The slowest shard of a split is the step of that
    #      line, so `speedup_bound` = N = 1 launch / slowest shard is what value(N) / value(1) can reach before xGMI, RCCL or a
    #      straggling GPU take anything away — measured here because a one-GPU box is the only hardware a build round gets.
    if stacks[0].layout == COLUMNS:
        base_ms = result["roofline"]["avg_launch_ms"]
        series = {}
        for parts in (2, 4, 8):
            cuts = plan.bounds(parts, target_cost=TARGET_COST_SHORT_LAUNCH)  # the cut the N-rank headline uses
            ms_each = []
            for r in range(parts):
                a, b = cuts[r], cuts[r + 1]
                i_r, w_r, rows_r = ordered_tables(idx64, w64, tgt_grid, a, b, np_dtype, dev, natural=rows_d is None)
                o_r = Stack.empty(b - a, n_lev, tdtype, dev, COLUMNS)
                fn = lambda: native.regrid_ell(stacks[0].data, o_r.data, i_r, w_r, n_src=n_src, n_tgt=b - a, k=args.k, n_lev=n_lev,  # noqa: E731
                                               src_pitch=stacks[0].pitch, out_pitch=o_r.pitch, layout=COLUMNS, tgt_rows=rows_r)
                ms_each.append(time_launches(fn, 20, 3)[0])
                del i_r, w_r, rows_r, o_r
            slowest = max(ms_each)
            series[str(parts)] = {"shard_ms": ms_each, "targets": [cuts[r + 1] - cuts[r] for r in range(parts)], "slowest_ms": slowest,
                                  "speedup_bound": base_ms / slowest, "value_bound": n_tgt * n_lev / (slowest * 1e-3)}
        extras["strong_scaling_shards_on_one_gpu"] = dict(series, n1_launch_ms=base_ms,
            target_cost=TARGET_COST_SHORT_LAUNCH,
            note="each rank's shard of the N-rank headline step timed alone on this GPU (HIP events, 20 launches); speedup_bound = n1_launch_ms / slowest shard")

    # the one-off index build on the device instead of cKDTree: raw kernel order, and with equidistant candidates settled by
    # cKDTree (the table the reference builds, bit for bit)
    args4 = (src_grid["latitudes"], src_grid["longitudes"], tgt_grid["latitudes"], tgt_grid["longitudes"])
    # first constructions, timed on the uncached workers (the tables of this grid pair are remembered — process and files — since
    # the precompute above; the host tree is forgotten first so that settling the ties pays its build)
    interp.knn_cache_clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    src_xyz, tgt_xyz = interp.unit_sphere_xyz(*args4[:2]), interp.unit_sphere_xyz(*args4[2:])
    raw_i = interp.device_knn(src_xyz, tgt_xyz, args.k, ties="index")[0]
    extras["knn_device_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    src_xyz, tgt_xyz = interp.unit_sphere_xyz(*args4[:2]), interp.unit_sphere_xyz(*args4[2:])
    di = interp.device_knn(src_xyz, tgt_xyz, args.k)[0]
    extras["knn_device_ties_settled_s"] = time.perf_counter() - t0
    del src_xyz, tgt_xyz
    interp.nearest_grid_points_device(*args4, num_neighbours_to_return=args.k)
    t0 = time.perf_counter()
    again = interp.nearest_grid_points_device(*args4, num_neighbours_to_return=args.k)
    extras["knn_remembered_s"] = time.perf_counter() - t0  # the same request again: process memo (interp._remembered_table)
    assert np.array_equal(again.reshape(n_tgt, -1), di)
    del again
    extras["knn_rows_identical_to_ckdtree"] = float((di.reshape(n_tgt, -1) == idx64).all(axis=1).mean())
    extras["knn_rows_identical_to_ckdtree_kernel_order"] = float((raw_i.reshape(n_tgt, -1) == idx64).all(axis=1).mean())
    del raw_i, di

    stacks.clear()
    outs.clear()
    torch.cuda.empty_cache()
    other = ("f32", torch.float32, np.float32, 4) if args.dtype == "f64" else ("f64", torch.float64, np.float64, 8)
    for name, dt, npdt, isz, lay in ((f"{other[0]}_columns", *other[1:], COLUMNS),
                                     (f"{args.dtype}_fields", tdtype, np_dtype, itemsize, FIELDS)):
        s = synth_stack(src_grid, n_lev, dt, dev, 0, lay)
        o = Stack.empty(n_tgt, n_lev, dt, dev, lay)
        # the other width in the headline's (ordered) traversal; the field-major kernel has no ordered form: natural tables
        i2, wd, r2 = ordered_tables(idx64, w64, tgt_grid, 0, n_tgt, npdt, dev, natural=lay != COLUMNS or args.natural_order)
        ms, _ = time_launches(lambda: native.regrid_ell(s.data, o.data, i2, wd, n_src=n_src, n_tgt=n_tgt, k=args.k, n_lev=n_lev, src_pitch=s.pitch,
                                                        out_pitch=o.pitch, layout=lay, tgt_rows=r2), 10, 2)
        extras[name] = line(n_tgt * n_lev, ms, algorithmic_bytes(n_lev, isz, n_unique, n_tgt, args.k))
        del i2, r2
        if name.endswith("_columns"):  # the k = 1 gather in the other width too
            ms, _ = time_launches(lambda: launch(s, o, idx=idx1, w=None, k=1, n_t=n_tgt), 10, 2)
            extras[f"nearest_k1_{other[0]}"] = line(n_tgt * n_lev, ms, algorithmic_bytes(n_lev, isz, int(np.unique(idx64[:, 0]).size), n_tgt, 1))
        del s, o, wd
        torch.cuda.empty_cache()
    if "cpu_baseline" in result:  # like-for-like ratios: the same statement in the same width on both sides (reported, not a target)
        cpu = result["cpu_baseline"]["variants"]
        gpu_k4 = {args.dtype: result["value"], other[0]: extras[f"{other[0]}_columns"]["value"]}
        gpu_k1 = {args.dtype: extras["nearest_k1"]["value"], other[0]: extras[f"nearest_k1_{other[0]}"]["value"]}
        extras["gpu_over_cpu_one_core"] = {f"k{args.k}_f64": gpu_k4["f64"] / cpu["csr_f64"]["value"], f"k{args.k}_f32": gpu_k4["f32"] / cpu["csr_f32"]["value"],
                                           "k1_f64": gpu_k1["f64"] / cpu["k1_f64"]["value"], "k1_f32": gpu_k1["f32"] / cpu["k1_f32"]["value"]}

    if args.headline_shape_only:
        result["extras"] = extras
        return

    # ---- BASELINE configs[1]: O96 -> 1 degree bilinear, ONE surface field (the thin-stack regime: a 1-level column stack)
    try:
        g96, g1 = lookup("o96"), lookup([1.0, 1.0])
        m = interp.bilinear_octahedral(96, g1)
        p2 = GatherPlan.from_matrix(m)
        u2 = int(np.unique(m["matrix_indices"]).size)
        c2 = {"workload": "O96 (40320 pts) -> 1 deg lat-lon (65160 pts), bilinear k=4 matrix, 1 field per launch", "calls": 200}
        tiny = torch.empty(1024, dtype=torch.float32, device=dev)
        tiny2 = torch.empty_like(tiny)
        floor = launch_times(lambda: native.stream_copy(tiny, tiny2), 200, 20)  # a 4 KB copy between two events: what ANY launch costs here
        c2["launch_floor_ms"] = {"avg": float(np.mean(floor)), "median": float(np.median(floor)), "min": float(np.min(floor)),
                                 "what": "atx_stream_copy of 4 KB between two HIP events, 200 calls"}
        for name, dt, isz in (("f64", torch.float64, 8), ("f32", torch.float32, 4)):
            s = synth_stack(g96, 1, dt, dev, 0, COLUMNS)
            alg2 = algorithmic_bytes(1, isz, u2, p2.n_tgt, 4)
            out2 = s.new_like(n_pts=p2.n_tgt)
            bound, _ = p2.bind(s, out2)
            routes = {"apply_allocating": lambda: p2.apply(s), "apply_out": lambda: p2.apply(s, out=out2), "bound": bound}
            rec = {}
            for route, fn in routes.items():
                ms = launch_times(fn, 200, 20)
                rec[route] = {"avg_launch_ms": float(np.mean(ms)), "median_launch_ms": float(np.median(ms)), "min_launch_ms": float(np.min(ms))}
            assert torch.equal(p2.apply(s).data, out2.data)  # the three routes launch the same kernel on the same tables
            best = rec["bound"]
            c2[name] = dict(line(p2.n_tgt, best["avg_launch_ms"], alg2), median_launch_ms=best["median_launch_ms"], min_launch_ms=best["min_launch_ms"],
                            route="GatherPlan.bind (arguments converted once, caller-kept output stack)", routes=rec,
                            avg_over_launch_floor=best["avg_launch_ms"] / c2["launch_floor_ms"]["avg"])
        c2["note"] = ("0.9 MB of algorithmic traffic per launch: launch-latency bound, not HBM bound; `routes` compares GatherPlan.apply (allocates its "
                      "output), apply(out=) and the bound call against the launch floor of the same box and run")
        extras["config2"] = c2
    except Exception as e:
        extras["config2"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- BASELINE configs[3]: O1280 -> N320-sized, 6 variables x 137 levels x 4 timesteps = 24 stacks, 8 target shards
    try:
        extras["config4"] = config4_lines(args, dev, torch.float32, np.float32, 4)  # float32 as SURVEY.md §8d sizes it (86.8 GB resident)
        if args.levels == 137:
            extras["config4"]["traffic"] = {
                "slowest_shard_24_stacks_of_137": carried_traffic("config4 o1280->n320-sized k=4 24x137 levels f32 columns shard 7 of 8"),
                "slowest_shard_4_stacks_of_822": carried_traffic("config4 o1280->n320-sized k=4 4x822 levels f32 columns shard 7 of 8"),
                "all_targets_4_stacks_of_822": carried_traffic("config4 o1280->n320-sized k=4 4x822 levels f32 columns all targets")}
    except Exception as e:
        extras["config4"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()

    # ---- BASELINE configs[4]: the chained filters on an ERA5-shape O2560 stack (14.4 GB), plain gather vs the fused launch
    try:
        plain, fused, units, _, alg5, keep = config5_case(args, dev, tdtype, np_dtype)
        ms_plain, _ = time_launches(plain, 10, 2)
        ms_fused, _ = time_launches(fused, 10, 2)
        extras["config5"] = {"workload": "O2560 (26306560 pts) -> 0.25 deg, k=4, 137 levels: regrid | orog_to_z (1 level) | convert K->degC (136 levels)",
                             "regrid_only": line(units, ms_plain, alg5), "fused_chain_one_launch": line(units, ms_fused, alg5)}
        del plain, fused, keep
        torch.cuda.empty_cache()
        # the same chain with the three variables of SURVEY.md §8d's description sharing a column: one stack of 3 x 137 levels
        plain, fused, units, _, alg5, keep = config5_case(args, dev, tdtype, np_dtype, variables=3)
        ms_plain, _ = time_launches(plain, 10, 2)
        ms_fused, _ = time_launches(fused, 10, 2)
        extras["config5"]["three_variables_share_a_column"] = {
            "workload": f"O2560 -> 0.25 deg, k=4, ONE stack of 3 x {args.levels} levels (t -> degC, orography-like x g, one variable left alone), "
                        f"{keep[0].data.numel() * keep[0].data.element_size() / 1e9:.1f} GB resident",
            "regrid_only": line(units, ms_plain, alg5), "fused_chain_one_launch": line(units, ms_fused, alg5)}
        del plain, fused, keep
        if args.levels == 137 and args.dtype == "f64":
            extras["config5"]["traffic"] = {
                "fused_137_levels": carried_traffic("config5 o2560->0.25 k=4 L=137 f64 columns fused regrid|orog_to_z|convert all targets"),
                "regrid_only_137_levels": carried_traffic("config5 o2560->0.25 k=4 L=137 f64 columns regrid only all targets"),
                "fused_3x137_levels": carried_traffic("config5 o2560->0.25 k=4 L=411 f64 columns fused regrid|orog_to_z|convert all targets")}
    except Exception as e:
        extras["config5"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()

    # the CPU's best case (SURVEY.md §8d baseline B): the same statement on every core this process may use (os.cpu_count() capped
    # by the cgroup quota — 16 on the MI355X boxes, where 32-256 workers measured SLOWER, profiles/r02_cpu_workers_sweep.jsonl), run
    # as a child process that never touches the GPU; reported next to the one-thread "as the reference runs" baseline
    if not args.no_cpu_baseline:
        import subprocess

        try:
            child = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_all_cores.py"), "--seconds", "8", "--workers", str(args.cpu_workers),
                                    "--src-grid", args.src_grid, "--tgt-grid", args.tgt_grid, "--k", str(args.k)],
                                   capture_output=True, text=True, timeout=240)
            extras["cpu_all_cores"] = json.loads(child.stdout.strip().splitlines()[-1])
        except Exception as e:  # a baseline must never take the bench line down
            extras["cpu_all_cores"] = {"error": f"{type(e).__name__}: {e}"}
    result["extras"] = extras


def config5_case(args, dev, tdtype, np_dtype, rank=0, world=1, variables=1):
    """BASELINE configs[4]: regrid + orography-adjust + unit-convert chained on ERA5-shape O2560 fields, as ONE fused launch per
    stack.  `variables=1` (rounds 1-3): one 137-level stack, 136 levels of t -> degC and one level of orog -> z.  `variables=3`
    (SURVEY.md §8d "137 levels x {t, orog-like, one convertible var}"): the three variables of a grid point share a column — 137
    levels of t (K -> degC), 137 of an orography-like field (x g), 137 left alone — one stack of 411 levels (43 / 87 GB), the layout
    a job that controls its stacks would choose (DESIGN.md §2 "Tall stacks").  Returns (launch closures, units, algorithmic bytes)
    for this rank's traffic-balanced slice of the 0.25 degree target grid.  The index table comes from the device k-NN search with
    the kernel's own order among equidistant points (cKDTree needs about a minute for 26.3 M points; the timing does not depend
    on that order)."""
    from anemoi_transform_amd import native
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.interp import knn_inverse_distance
    from anemoi_transform_amd.stack import COLUMNS, Stack

    g_src, g_tgt = lookup("o2560"), lookup("0.25")
    n5_src, n5_tgt, L1 = len(g_src["latitudes"]), len(g_tgt["latitudes"]), args.levels
    L = L1 * variables
    idx5, w5 = knn_inverse_distance(g_src, g_tgt, k=4, device=True, ties="index")
    b5 = GatherPlan(n5_src, n5_tgt, index=idx5, weights=w5).bounds(world)
    lo, hi = b5[rank], b5[rank + 1]
    if variables == 1:
        x = synth_stack(g_src, L, tdtype, dev, 0, COLUMNS)
    else:
        x = Stack.empty(n5_src, L, tdtype, dev, COLUMNS, zero=True)
        gen = torch.Generator(device=dev)
        gen.manual_seed(SEED + 5)
        x.data[:, :L].normal_(270.0, 15.0, generator=gen)
    out = Stack.empty(hi - lo, L, tdtype, dev, COLUMNS)
    idx_d, w_d, rows_d = ordered_tables(idx5, w5, g_tgt, lo, hi, np_dtype, dev, natural=args.natural_order)
    cp, to_z, to_degc = (native.OP_COPY, 0, 0.0, 0.0), (native.OP_MUL, 0, 9.80665, 0.0), (native.OP_AFFINE, 0, 1.0, -273.15)
    if variables == 1:
        stages = [[cp] * (L - 1) + [to_z], [to_degc] * (L - 1) + [cp]]
    else:  # levels [0, L1): t; [L1, 2 L1): orography-like; the rest: left alone
        stages = [[cp] * L1 + [to_z] * L1 + [cp] * (L - 2 * L1), [to_degc] * L1 + [cp] * (L - L1)]
    prog = native.level_program(stages, dev)
    kw = dict(n_src=n5_src, n_tgt=hi - lo, k=4, n_lev=L, src_pitch=x.pitch, out_pitch=out.pitch, layout=COLUMNS, tgt_rows=rows_d)
    plain = lambda: native.regrid_ell(x.data, out.data, idx_d, w_d, **kw)  # noqa: E731
    fused = lambda: native.regrid_ell(x.data, out.data, idx_d, w_d, prog=prog, n_stage=2, **kw)  # noqa: E731
    itemsize = 4 if tdtype == torch.float32 else 8
    alg = algorithmic_bytes(L, itemsize, int(np.unique(idx5[lo:hi]).size), hi - lo, 4)
    return plain, fused, (hi - lo) * L, n5_tgt * L, alg, (x, out, idx_d, w_d, rows_d, prog)


def config4_lines(args, dev, tdtype, np_dtype, itemsize):
    """The whole 3 288-field batch of BASELINE configs[3] resident on ONE MI355X (88.7 GB f32): all target points (what one
    GPU does alone), and each of the 8 traffic-balanced target shards (what each of 8 GPUs would do; the slowest bounds the job) —
    in two layouts of the same fields: 24 stacks of 137 levels (one per variable and timestep) through the batched launch, and
    4 stacks of 822 levels (one per timestep: the 6 variables of a grid point share a column of 3.3 KB), where a gathered column
    wastes less of its first and last 128-byte line (tools/experiments/tall_stacks.py: 0.69 -> 0.73 of the HBM peak)."""
    from anemoi_transform_amd import native
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.interp import knn_inverse_distance
    from anemoi_transform_amd.stack import COLUMNS, Stack

    src_grid, tgt = lookup("o1280"), lookup("n320-sized")  # BASELINE configs[3], whatever grid pair the headline runs on
    n_src, n_tgt, n_var, n_time, k = len(src_grid["latitudes"]), len(tgt["latitudes"]), 6, 4, 4
    idx, w = knn_inverse_distance(src_grid, tgt, k=k)
    plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
    bounds = plan.bounds(8)
    n_fields = n_var * n_time * args.levels

    def layout(n_stack, n_lev):
        gen = torch.Generator(device=dev)
        stacks = []
        for i in range(n_stack):
            gen.manual_seed(SEED + 7 * i)
            st = Stack.empty(n_src, n_lev, tdtype, dev, COLUMNS, zero=True)
            st.data[:, :n_lev].normal_(250.0 + 5.0 * (i // 4), 20.0, generator=gen)
            stacks.append(st)
        outs = [Stack.empty(n_tgt, n_lev, tdtype, dev, COLUMNS) for _ in range(n_stack)]

        def run(lo, hi):
            idx_d, w_d, rows_d = ordered_tables(idx, w, tgt, lo, hi, np_dtype, dev, natural=args.natural_order)
            views = [o.data[lo:hi] for o in outs]

            def go():  # atx_regrid_ell_batch / _ordered: one launch per 16 stacks
                native.regrid_ell_batch([s.data for s in stacks], views, idx_d, w_d, n_src=n_src, n_tgt=hi - lo, k=k, n_lev=n_lev,
                                        src_pitch=stacks[0].pitch, out_pitch=outs[0].pitch, layout=COLUMNS, tgt_rows=rows_d)

            ms, _ = time_launches(go, 10, 2)
            u = int(np.unique(idx[lo:hi]).size)
            return dict(line((hi - lo) * n_lev * n_stack, ms, n_stack * algorithmic_bytes(n_lev, itemsize, u, hi - lo, k)), targets=hi - lo)

        rec = {"stacks": n_stack, "levels_per_stack": n_lev, "resident_GB": sum(s.data.numel() for s in stacks) * itemsize / 1e9,
               "all_targets_one_gpu": run(0, n_tgt)}
        shards = [run(bounds[r], bounds[r + 1]) for r in range(8)]
        slowest = max(shards, key=lambda s: s["avg_launch_ms"])
        rec["shards_of_8"] = {"ms": [s["avg_launch_ms"] for s in shards], "targets": [s["targets"] for s in shards], "slowest": slowest,
                              "job_value_bound_by_slowest_shard": n_tgt * n_fields / (slowest["avg_launch_ms"] * 1e-3),
                              "note": "each shard timed alone on this GPU with all source stacks resident: the per-GPU step of the 8-GPU job"}
        del stacks, outs
        torch.cuda.empty_cache()
        return rec

    per_variable = layout(n_var * n_time, args.levels)
    per_timestep = layout(n_time, n_var * args.levels)
    out = {"dtype": "f32" if itemsize == 4 else "f64",
           "workload": f"O1280 -> N320-sized reduced Gaussian ({n_tgt} pts, grids.sized_row_lengths), k=4, {n_var} variables x {args.levels} levels x "
                       f"{n_time} timesteps = {n_fields} fields resident ({per_variable['resident_GB']:.1f} GB)",
           # round 1-3's keys keep their meaning: the 24 x 137 layout
           "all_targets_one_gpu": per_variable["all_targets_one_gpu"], "shards_of_8": per_variable["shards_of_8"],
           "stacks_per_variable_and_timestep": per_variable,
           "stacks_per_timestep_variables_share_a_column": per_timestep}
    return out


if __name__ == "__main__":
    main()
