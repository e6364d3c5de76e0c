#!/usr/bin/env python3
"""bench.py -- BAM-record decode -> CIGAR walk -> pileup on MI355X, BASELINE.json's metric.

One "step" = one full pass of the hot path over one synthetic coordinate-sorted BAM record stream that
is already resident in HBM: record framing (K0), fixed fields + end_position (K1), SEQ/QUAL/CIGAR
unpack (K2), reference-run keys (K3) and the per-position pileup (K4: plan + accumulate).  Nothing is cached
between steps (bgp_reset drops every result).

N GPUs (BASELINE.json config 4): the SAME 100 M-read BAM split by reference contig over the ranks (strong
scaling; `--weak` gives every rank a full-size shard instead), no data-path collective; the path's only
exchange is the whole-BAM depth histogram, summed inside the timed step by the product's own
bgp_hist_reduce on an RCCL communicator this script creates (ncclGetUniqueId -> ncclCommInitRank).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8                     # spawns its 8 ranks itself when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
STAGES = {"K012 decode in one pass": ("k012_", "k01_"), "K0 framing": ("k0_",), "K1 record table": ("k1_",), "K2 unpack": ("k2_", "k12_"), "K3 run keys": ("k3_",),
          "K4 pileup columns": ("k4_",)}


def kernels_sha():
    """Hash of the kernel sources: a committed PMC profile is only quoted while it describes these kernels."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "biod_amd", "csrc", "*"))):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# kernels behind the timers of a step (a timer can bracket several launches; kernels of the set-up, e.g. the flush before the
# timed region, are not the step's)
TIMER_KERNELS = {"k12_lds": ("k12_lds",), "k012_stream": ("k_stream_decode",), "k2_unpack": ("k2_unpack",), "k12_fused": ("k12_fused",), "k1_rows": ("k12_fused",),
                 "k4_bounds": ("k4_bounds",), "k4_plan": ("k4_plan(", "k4_wscan"), "k4_plan_rest": ("k4_plan_rest", "k4_wscan"), "k4_acc2": ("k4_acc",), "k4_tile": ("k4_tile",), "k4_stats": ("k4_stats",),
                 "k4_checkpoints": ("k4_ck_",),
                 "k0_guess": ("k0_guess",), "k0_chase": ("k0_chase", "k0_verify"), "k0_rechase": ("k0_rechase",), "k3_keys_scan": ("k3_",)}


def pmc_traffic(timers, config, n_reads, launches=None):
    """HBM bytes per step of the kernels behind the timers of the stage: the mean per dispatch of each kernel times the
    timer's launches per step, from the committed
    rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, FETCH doubled for
    gfx950 as MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside this process, so the number is only
    reported when the workload is the one the profile was taken on (configuration and read count in the profile's header) AND
    that header carries the hash of the kernel sources as they are now (scratch/refresh_profiles.sh writes it); otherwise null."""
    path = os.path.join(ROOT, "profiles", f"r06_pmc_hbm_traffic_{config}.csv")
    if not os.path.exists(path):
        return None
    lines = open(path).read().splitlines()
    if not any(("kernels_sha=" + kernels_sha()) in ln for ln in lines if ln.startswith("#")) or not any(f"reads={n_reads}" in ln for ln in lines if ln.startswith("#")):
        return None
    total = None
    for t in timers:
        names = tuple(TIMER_KERNELS.get(t, ()))
        for line in lines:
            f = line.strip().split(",")
            if len(f) >= 5 and names and f[0].startswith(names) and f[4].isdigit():
                per_dispatch = int(f[4])                    # (2 x FETCH + WRITE -- for the chase's sparse 4-byte hops too: profiles/r03_pmc_chase_probe.txt)
                total = (total or 0) + per_dispatch * int((launches or {}).get(t, 1))
    return total


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg5"], help="BASELINE.json config (cfg3 = the 100M-read mixed-CIGAR BAM the metric is quoted on)")
    ap.add_argument("--reads", type=int, default=0, help="reads of the whole BAM (0 = the config's own size, reduced only if HBM is short)")
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="reads in the CPU-baseline sample (0 = auto, about 10-30 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-checks", action="store_true", help="skip the full-size invariants after the timed region")
    ap.add_argument("--parity", action="store_true", help="the oracle's interior-shard check even with --no-cpu-baseline (it is on by default otherwise)")
    ap.add_argument("--no-md", action="store_true", help="skip the use_md_tag leg (reference_base from MD tags, what examples/make_pileup.d:16 asks for) after the timed region")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end legs after the timed region (device-E2E from host memory through batched staging; file-E2E from a BGZF file through the device inflate)")
    ap.add_argument("--e2e-reads", type=int, default=0, help="reads of the end-to-end legs' input (0 = 10 M, or the workload's own size if that is smaller)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: every rank gets a full --reads-sized shard of its own contigs (default: strong, ONE BAM split by contig)")
    ap.add_argument("--strong", action="store_true", help="(default) kept for older command lines")
    return ap.parse_args()


def usable_cores():
    """Host cores this process can really run on: the scheduler's affinity mask, cut by the cgroup CPU quota
    (the GPU boxes show 256 logical CPUs under a 16-CPU cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(cfg_name, n_sample):
    """The oracle (a C restatement of BioD's algorithm -- BioD itself is D and cannot be built here)
    timed on this box's host cores on a bounded sample of the same workload: median of three runs."""
    import oracle_lib as O
    from biod_amd.synth import host_stream_np, scaled
    cores = usable_cores()
    if n_sample <= 0:
        n_probe = 20_000 if cfg_name != "cfg5" else 60
        probe = host_stream_np(scaled(cfg_name, n_probe))
        t0 = time.perf_counter(); O.pileup_bench(probe, 1); dt = max(time.perf_counter() - t0, 1e-4)
        n_sample = int(min(max(8.0 / (dt / n_probe), n_probe), 3_000_000))      # ~8 CPU-seconds of pileup work per run, three runs
        if cfg_name != "cfg5":
            # every thread needs work: pileupChunks cuts 16 384 000-byte chunks (~56k reads); and 10 M reads (a tenth of cfg3,
            # ~3 s a run on 16 cores) keep the run-to-run spread of a 3 M-read sample out of the figure
            n_sample = int(min(max(n_sample, cores * 2 * 56_000, 10_000_000), 32_000_000))
    stream = host_stream_np(scaled(cfg_name, n_sample))
    rates, rp = [], None
    for _ in range(3):
        t0 = time.perf_counter(); rp = O.pileup_bench(stream, cores); tp = time.perf_counter() - t0
        assert rp.status in (0, -9), rp.status
        rates.append(n_sample / tp / 1e6)
    rates.sort()
    out = {"value": round(rates[1], 4), "unit": "Mreads/s", "cores": cores, "kind": "port", "runs": 3,
           "min": round(rates[0], 4), "max": round(rates[2], 4),
           "columns_per_s": round(rp.n_columns * rates[1] * 1e6 / n_sample, 1),
           "sample": f"{n_sample} reads of {cfg_name}'s read model at its coverage depth ({len(stream) / 1e6:.0f} MB inflated), "
                     f"pileupChunks-style decomposition over {cores} threads (= usable cores: {os.cpu_count()} logical CPUs, affinity and cgroup quota applied), "
                     f"consumer touches coverage/n_starting_here/every base+quality; median of 3 runs"}
    # the serial sweep (one thread, BioD's pileupColumns as it is written) beside the threaded decomposition
    n_ser = min(n_sample, 1_000_000 if cfg_name != "cfg5" else n_sample)
    ser = stream if n_ser == n_sample else host_stream_np(scaled(cfg_name, n_ser))
    t0 = time.perf_counter(); O.pileup_bench(ser, 1); ts = time.perf_counter() - t0
    out["serial_value"] = round(n_ser / ts / 1e6, 4)
    out["serial_sample_reads"] = n_ser
    return out


def oracle_interior_check(ctx, cfg, cfg_name, info, torch, dev):
    """Interior-shard parity on the FULL-size result the timed steps left on the device: >= 16 record ranges (the first
    and last records, ranges across 4 / 8 / 16 / 24 GiB of stream offset, across 4 / 8 GiB of the arenas and of the column
    array, across contig boundaries, at the deepest column) regenerated on the host, run through the oracle and compared
    bit for bit -- rows, stream bytes, SEQ / QUAL / CIGAR slices, every column the range determines, per-read entries on
    two of them (tests/interior_parity.py)."""
    import interior_parity as IP
    return IP.run(ctx, cfg, info, 20_000 if cfg_name != "cfg5" else 200, torch, dev, want=16)


def md_leg(ctx, cfg, cfg_name, info, n_reads, reps=3):
    """`use_md_tag = true` on the resident result (examples/make_pileup.d:16 calls makePileup(bam.reads, true): PileupRangeUsingMdTag,
    pileup.d:522-654): K6's kernels timed by their HIP events over `reps` runs of bgp_pileup_begin(use_md_tag = 1) -- K3 / K4 redone in
    front of each, untimed here, so that nothing of K6 is cached --, the wall time up to the first window, and the oracle's sweep over
    the stream's first records against the device's bytes (tests/interior_parity.py: md_prefix_check).
    Algorithmic bytes: the 32-byte rows in, one byte per position out (the reference looks at a PROVIDER's MD tag, SEQ and CIGAR
    only -- pileup.d:560-562 -- and so does K6: those bytes, a few per cent of the reads', are left out of the figure)."""
    from biod_amd import _lib as L
    import interior_parity as IP
    ks, walls = [], []
    for _ in range(reps):
        ctx.pileup_run()
        ctx.timing_enable(True); ctx.timing_reset()
        t0 = time.perf_counter()
        gen = ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=True, skip_zero_coverage=True, window_positions=1 << 20, copy=False)      # (views of the ctx's pinned window: the facade's copy into fresh arrays was 16 of the 26 ms this reported)
        next(gen)
        walls.append((time.perf_counter() - t0) * 1e3)
        gen.close()
        ks.append({k: v["total_ms"] for k, v in ctx.timing_get().items() if k.startswith("k6") and v["launches"]})
        ctx.timing_enable(False)
    names = sorted({k for d in ks for k in d})
    med = {k: sorted(d.get(k, 0.0) for d in ks)[len(ks) // 2] for k in names}
    k6_ms = sum(med.values())
    alg = 32 * n_reads + int(info.n_positions)
    out = {"what": "reference_base from MD tags for every position of the piled-up BAM (bgp_pileup_begin, use_md_tag = 1): K6 = " + " + ".join(names),
           "k6_ms": round(k6_ms, 3), "kernels_ms": {k: round(v, 3) for k, v in med.items()}, "wall_ms_to_first_window": round(sorted(walls)[len(walls) // 2], 2),
           "algorithmic_bytes": alg, "algorithmic_GBps": round(alg / (k6_ms * 1e-3) / 1e9, 1) if k6_ms else None,
           "frac": round(alg / (k6_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if k6_ms else None, "reps": reps}
    try:
        out["prefix_vs_oracle"] = dict(IP.md_prefix_check(ctx, cfg, 20_000 if cfg_name != "cfg5" else 200), equal=True)
        # ... and at the start of every contig (provider chains restart at a reference change, pileup.d:573-579): records 12.5 M, 25 M, ... of cfg3
        per = IP.md_contig_prefix_check(ctx, cfg, 20_000 if cfg_name != "cfg5" else 200)
        out["prefix_vs_oracle"]["contigs"] = per
        out["prefix_vs_oracle"]["ranges"] = len(per)
    except AssertionError as e:
        out["prefix_vs_oracle"] = {"equal": False, "why": str(e)[:300]}
    if cfg_name in ("cfg2", "cfg5"):      # streams without N skips: every covered column of the WHOLE result against the generator's own reference (a property, not the oracle)
        try:
            out["whole_stream_vs_generator_reference"] = IP.md_generator_reference_check(ctx, cfg)
        except AssertionError as e:
            out["whole_stream_vs_generator_reference"] = {"equal": False, "why": str(e)[:300]}
    return out


def entries_leg(ctx, cfg_name, n_reads, sum_lseq, sum_ncig, want_positions=4_000_000):
    """K5, the rest of a PileupColumn (column.reads: PileupRead's cursor per read and column, pileup.d:86-230,267-274): windows WITH
    entries over the first `want_positions` positions of the resident result, K5's kernels timed by their HIP events.  Algorithmic bytes:
    16 per entry written, plus what the reads that make them are (row, CIGAR, packed SEQ, QUAL: the stream-wide average per read times the
    reads a position's share of the stream amounts to).  The windows' copy to the host is in the wall time only."""
    import ctypes as C
    from biod_amd import _lib as L
    win = int(os.environ.get("BGP_BENCH_ENTRY_WINDOW", 1 << 22))      # positions a window (the consumer's choice: bgp_pileup_begin)
    walls = []
    for _rep in range(2):      # (twice: the first call also pins the host buffer the entries land in -- gigabytes, 12 GB/s of hipHostMalloc)
        ctx.timing_enable(True); ctx.timing_reset()
        L.check(L.lib().bgp_pileup_begin(ctx.handle, L.BGP_PILEUP_COLUMNS, 0, 0, (1 << 64) - 1, 1, 1, win), ctx.handle)
        t0 = time.perf_counter(); n_pos = 0; n_ent = 0
        cw = L.Columns()
        while n_pos < want_positions:
            rc = L.check(L.lib().bgp_pileup_next_window(ctx.handle, C.byref(cw)), ctx.handle)
            if rc == L.BGP_EOF:
                break
            n_pos += cw.n_positions
            n_ent += C.cast(cw.entry_off, C.POINTER(C.c_uint64))[cw.n_positions]
        walls.append((time.perf_counter() - t0) * 1e3)
    wall = walls[-1]
    tm = {k: v["total_ms"] for k, v in ctx.timing_get().items() if k.startswith("k5") and v["launches"]}
    ctx.timing_enable(False)
    k5_ms = sum(tm.values())
    info_pos = ctx.pileup_info().n_positions if hasattr(ctx, "pileup_info") else None
    per_read = 32 + (4 * sum_ncig + (sum_lseq + 1) // 2 + sum_lseq) / max(n_reads, 1)
    return {"what": "per-read entries (column.reads) of the first windows of the piled-up BAM: K5 = " + " + ".join(sorted(tm)), "positions": n_pos, "window_positions": win, "entries": int(n_ent),
            "k5_ms": round(k5_ms, 3), "kernels_ms": {k: round(v, 3) for k, v in tm.items()}, "G_entries_per_s_kernel": round(n_ent / (k5_ms * 1e-3) / 1e9, 2) if k5_ms else None,
            "wall_ms_with_host_copies": round(wall, 1), "wall_ms_first_call": round(walls[0], 1), "bytes_per_read": round(per_read, 1), "_n_pos_total": info_pos}


from biod_amd.checks import DevArray as _DevArray, full_size_invariants      # device-side identities of the result (no oracle)


def rccl_libraries_mapped():
    """Paths of the librccl objects mapped into this process (/proc/self/maps)."""
    seen = []
    try:
        for ln in open("/proc/self/maps"):
            f = ln.split()
            if len(f) >= 6 and "librccl" in os.path.basename(f[5]) and f[5] not in seen:
                seen.append(f[5])
    except OSError:
        pass
    return seen


def rccl_communicator(torch, dist, rank, world, dev):
    """An RCCL communicator of this script's own (the product takes it as a void*): rank 0's unique id goes round
    through torch.distributed, every rank joins with ncclCommInitRank."""
    # ONE RCCL in the process: torch.distributed has mapped its own copy by now (torch/lib/librccl.so) -- this script's communicator and the
    # product's ncclAllReduce (api.cpp: bgp_hist_reduce takes whichever librccl is mapped already) come from that very object
    mapped = rccl_libraries_mapped()
    try:
        rccl = C.CDLL(mapped[0] if mapped else "librccl.so.1", mode=C.RTLD_GLOBAL)
    except OSError:
        rccl = C.CDLL("librccl.so", mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    if rank == 0:
        assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=dev)
    dist.broadcast(t, 0)
    C.memmove(C.byref(uid), bytes(t.cpu().tolist()), 128)
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rc = rccl.ncclCommInitRank(C.byref(comm), world, uid, rank)
    if rc != 0:
        raise RuntimeError(f"ncclCommInitRank failed: {rc}")
    return rccl, comm


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: become the launcher."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a.gpus)
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    # The result is ONE line on stdout.  Native libraries print there too (RCCL's version banner goes to the C stdout, which is
    # flushed at exit -- after the JSON line): file descriptor 1 is pointed at stderr for everything but that line.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s); start it as `python bench.py --gpus N` "
                 f"or under torch.distributed.run with --nproc-per-node equal to --gpus")
    import numpy as np
    import torch
    import torch.distributed as dist
    from biod_amd import _lib as L
    from biod_amd.pileup import Context
    from biod_amd.synth import CONFIGS, scaled

    if torch.cuda.device_count() and local >= torch.cuda.device_count():
        local %= torch.cuda.device_count()          # (only for dry runs of the N>1 path on a box with fewer GPUs)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl = comm = None
    with_comm = world > 1 or bool(os.environ.get("BGP_BENCH_FORCE_COMM"))    # (FORCE_COMM: the N > 1 code path on a one-GPU box, one-rank communicator)
    # REHEARSAL of N > 1 on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device): the ranks share the device, every
    # rank runs the real kernels on its own shard, and this SCRIPT sums a host copy of the depth histogram over gloo in place of the
    # product's bgp_hist_reduce.  Never a measurement of config 4 -- the line says so.
    reduce_gloo = with_comm and (os.environ.get("BGP_BENCH_REDUCE", "") == "gloo" or (world > 1 and torch.cuda.device_count() < world))
    if with_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s_.getsockname()[1]); s_.close()
        if reduce_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)      # nccl = RCCL over xGMI
            rccl, comm = rccl_communicator(torch, dist, rank, world, dev)

    def allred(x, op=None):
        """all-reduce of a device tensor over the process group (through the host when the group is gloo)"""
        op = op or dist.ReduceOp.SUM
        if reduce_gloo:
            y = x.cpu(); dist.all_reduce(y, op=op); x.copy_(y)
        else:
            dist.all_reduce(x, op=op)

    total_reads = a.reads or CONFIGS[a.config]["n_reads"]
    from biod_amd.shard import strong_cfg, weak_cfg
    strong = not a.weak
    free, _total = torch.cuda.mem_get_info(dev)
    need_per_read = 1000 if a.config != "cfg5" else 600_000       # stream + tables + arenas + columns + slabs, generous
    per_rank = total_reads if a.weak else -(-total_reads // world)
    while per_rank * need_per_read > free * 0.9 and total_reads > 1000:
        total_reads //= 2; per_rank //= 2
    if strong:          # BASELINE.json config 4 literally: the same BAM split by contig over the ranks
        cfg = strong_cfg(a.config, total_reads, rank, world)
    else:               # every rank owns a full-size shard of its own contigs
        cfg = weak_cfg(a.config, total_reads, rank)
    n_reads = int(cfg.n_reads)

    ctx = Context(local)
    t0 = time.perf_counter()
    stream_bytes = ctx.synth_device(cfg)                           # synthetic stream generated straight into HBM
    ctx.flush()
    r = ctx.decode_device(0)
    sptr, slen = r.stream, r.stream_len
    sum_lseq = int(torch.as_tensor(_DevArray(r.seq_off, (int(r.n) + 1,), "<i8"), device=dev)[-1].item())
    sum_ncig = int(torch.as_tensor(_DevArray(r.cigar_off, (int(r.n) + 1,), "<i8"), device=dev)[-1].item())
    gen_s = time.perf_counter() - t0

    def step():
        ctx.reset()
        ctx.attach_device_stream(sptr, slen)
        ctx.flush(decode=L.BGP_FETCH_ALL)                          # K0 framing, then record table + seq/qual/cigar arenas in one kernel over LDS windows (BGP_LDS_DECODE=0: k1_rows, k2_unpack)
        ctx.decode_device(L.BGP_FETCH_ALL)                         # (the arenas are there already: hands out the pointers)
        info = ctx.pileup_run()                                    # K3 + K4 dense columns
        if with_comm and reduce_gloo:                              # (rehearsal: the same exchange by this script, through the host)
            allred(torch.as_tensor(_DevArray(info.device_depth_hist, (L.BGP_DEPTH_BINS,), "<i8"), device=dev))
        elif with_comm:                                            # the path's only exchange: whole-BAM depth histogram, RCCL over xGMI, in place
            L.check(L.lib().bgp_hist_reduce(ctx.handle, comm, C.c_void_p(info.device_depth_hist), L.BGP_DEPTH_BINS), ctx.handle)
        return info

    def sync():
        torch.cuda.synchronize(dev)
        if with_comm:
            dist.barrier()

    for _ in range(a.warmup):
        info = step()
    ctx.timing_enable(True); ctx.timing_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        info = step()
    sync()
    dt = time.perf_counter() - t0
    timers = ctx.timing_get()
    ctx.timing_enable(False)

    # ---- outside the timed region: what the ranks produced, checked
    checks = None
    if not a.no_checks:
        checks = full_size_invariants(torch, dev, ctx, info, n_reads, own_histogram=not (with_comm and world > 1))      # (N > 1: the histogram on the device is the job's by now, checked against the job's totals below)
    # the oracle on this rank's own shard (records are functions of (contig, index): every rank regenerates its ranges on the host)
    parity = None
    if not a.no_cpu_baseline or a.parity:
        parity = oracle_interior_check(ctx, cfg, a.config, info, torch, dev)
    hist_ok = None
    if with_comm:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        allred(t, dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([info.n_columns, info.sum_coverage, n_reads, int(bool(checks is None or checks["ok"])), int(info.n_positions)], dtype=torch.int64, device=dev)
        allred(tot)
        job_columns, job_cov, job_reads, n_ok, job_positions = (int(x) for x in tot.tolist())
        # the reduced histogram the product left behind == the sum of the ranks' own histograms (recomputed, summed by torch)
        reduced = torch.as_tensor(_DevArray(info.device_depth_hist, (L.BGP_DEPTH_BINS,), "<i8"), device=dev).clone()
        ctx.reset(); ctx.attach_device_stream(sptr, slen); ctx.flush()
        mine = torch.as_tensor(_DevArray(ctx.pileup_run().device_depth_hist, (L.BGP_DEPTH_BINS,), "<i8"), device=dev).clone()
        allred(mine)
        hist_ok = bool(torch.equal(mine, reduced)) and int(reduced.sum().item()) > 0
        if int(info.max_coverage) < L.BGP_DEPTH_BINS - 1:      # the job's histogram against the job's totals: a position per count, the counts' weighted sum = the summed coverage
            hist_ok = hist_ok and int(reduced.sum().item()) == job_positions and int((reduced * torch.arange(L.BGP_DEPTH_BINS, device=dev)).sum().item()) == job_cov
        if checks is not None:
            checks["ok_on_ranks"] = n_ok
            checks["ok"] = n_ok == world
        if parity is not None:
            pt = torch.tensor([int(parity["all_equal"]), parity["ranges"], parity["reads"], parity["positions"], parity["entries"]], dtype=torch.int64, device=dev)
            allred(pt)
            parity["all_equal_on_ranks"] = int(pt[0].item()); parity["all_equal"] = int(pt[0].item()) == world
            parity["ranges"], parity["reads"], parity["positions"], parity["entries"] = (int(x) for x in pt[1:].tolist())
            parity["checked"] = [f"(rank 0 of {world}) " + c_ for c_ in parity["checked"]]
    else:
        job_columns, job_cov, job_reads = info.n_columns, info.sum_coverage, n_reads

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        kt = {k: v for k, v in timers.items() if v["launches"]}
        per_kernel = {k: {"ms_per_launch": round(v["total_ms"] / v["launches"], 4), "launches_per_step": v["launches"] // a.steps,
                          "algorithmic_GBps": round(v["algorithmic_bytes"] / v["launches"] / (v["total_ms"] / v["launches"] * 1e-3) / 1e9, 1) if v["total_ms"] > 0 else None}
                      for k, v in kt.items()}
        lps = {k: v["launches_per_step"] for k, v in per_kernel.items()}
        # a stage = the kernels that together produce one product of the path; its roofline figure is the stage's
        # algorithmic bytes over ALL of its kernels' time
        stages = {}
        for name, pre in STAGES.items():
            ks = [k for k in kt if k.startswith(pre)]
            if ks:
                stages[name] = {"ms_per_step": sum(kt[k]["total_ms"] for k in ks) / a.steps, "kernels": ks, "prefixes": pre}
        for name, st in stages.items():
            st["bytes"] = sum(kt[k]["algorithmic_bytes"] for k in st["kernels"]) // a.steps
        if "k4_acc2" in kt or "k4_tile" in kt:      # K4's algorithmic bytes (row + CIGAR + packed SEQ per read, 32 B per position) are booked on the accumulate launch (k4_acc2; on the general kernel, k4_tile, where that takes every window); the plan's own reads are not algorithmic
            stages["K4 pileup columns"]["bytes"] = sum(kt[k]["algorithmic_bytes"] for k in ("k4_acc2", "k4_tile") if k in kt) // a.steps
        dec_bytes = stream_bytes + 32 * n_reads + 2 * sum_lseq + 4 * sum_ncig      # whole record in, row + seq + qual + cigar out
        dom = max(stages, key=lambda s: stages[s]["ms_per_step"])
        dom_bytes = stages[dom]["bytes"]
        achieved = dom_bytes / (stages[dom]["ms_per_step"] * 1e-3) / 1e9
        dec_ms = sum(stages[s]["ms_per_step"] for s in stages if s[:2] in ("K0", "K1", "K2"))
        out = {
            "metric": "Mreads/s decoded + piled up (BAM record decode -> CIGAR walk -> per-base pileup, input resident in HBM)",
            "value": round(job_reads / dt * a.steps / 1e6, 3), "unit": "Mreads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8/u32 integer", "data": "synthetic",
            "config": {"workload": f"{a.config}: {job_reads if strong else n_reads} reads{'' if strong else '/GPU'} synthetic coordinate-sorted BAM record stream "
                                   f"({'mixed M/I/D/S/N CIGAR' if cfg.profile == 1 else 'M-only CIGAR' if cfg.profile == 0 else 'long indel-heavy reads'}, "
                                   f"l_seq {cfg.len_min}-{cfg.len_max}, {cfg.n_ref * (world if strong else 1)} contigs x {cfg.ref_len} bp"
                                   f"{', split by contig over the ranks' if strong and world > 1 else ''})",
                       "reads_per_gpu": n_reads, "inflated_bytes_per_gpu": stream_bytes, "parallelism": f"contig-sharded x{world}",
                       "step": "K0 framing + K1 record table and K2 seq/qual/cigar unpack (one kernel over LDS windows, k12_lds) + K3 run keys + K4 pileup columns (plan + accumulate)"
                               + (" + depth-histogram all-reduce (bgp_hist_reduce, RCCL)" if with_comm else "")},
            "pileup_columns_per_s": round(job_columns / dt * a.steps, 1),
            "columns_per_step": job_columns, "sum_coverage_per_step": job_cov,
            "decode_Mreads_per_s_kernel_only": round(n_reads / (dec_ms * 1e-3) / 1e6, 2) if dec_ms else None,
            "roofline": {"bound": "hbm", "kernel": dom + " = " + " + ".join(stages[dom]["kernels"]), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_launch": dom_bytes,
                         "traffic": pmc_traffic(stages[dom]["kernels"], a.config, n_reads, lps)},
            "roofline_decode_8d": {"what": "K0 + K1 + K2 by SURVEY 8(d)'s bytes: whole record in, 32-byte row + seq + qual + cigar out", "ms": round(dec_ms, 3),
                                   "achieved": round(dec_bytes / (dec_ms * 1e-3) / 1e9, 1) if dec_ms else None, "frac": round(dec_bytes / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if dec_ms else None},
            "stages_ms": {s: round(v["ms_per_step"], 3) for s, v in stages.items()},
            "stage_rooflines": {s: {"algorithmic_GBps": round(v["bytes"] / (v["ms_per_step"] * 1e-3) / 1e9, 1), "frac": round(v["bytes"] / (v["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                    "algorithmic_bytes": v["bytes"], "traffic": pmc_traffic(v["kernels"], a.config, n_reads, lps)}
                                for s, v in stages.items() if v["bytes"] and v["ms_per_step"] > 0},
            "kernels": per_kernel,
            "setup_s": round(gen_s, 2),
        }
        if checks is not None:
            out["checks_full_size"] = checks
        if hist_ok is not None:
            out["hist_reduce_equals_sum_of_ranks"] = hist_ok
            out["reduce"] = ("REHEARSAL: %d ranks on %d GPU(s), the histogram summed by bench.py over gloo through the host -- not a measurement of config 4" % (world, torch.cuda.device_count())) if reduce_gloo \
                else "bgp_hist_reduce: ncclAllReduce on a communicator of this script's making"
            out["rccl_libraries_mapped"] = rccl_libraries_mapped()
        if parity is not None:
            out["parity_interior"] = parity
        if world == 1 and not a.no_md:
            out["md_path"] = md_leg(ctx, cfg, a.config, info, n_reads)
            e = entries_leg(ctx, a.config, n_reads, sum_lseq, sum_ncig)
            e.pop("_n_pos_total", None)
            reads_share = n_reads * e["positions"] / max(int(info.n_positions), 1)          # the reads a stretch of that many positions holds, on average
            e["algorithmic_bytes"] = int(16 * e["entries"] + reads_share * e.pop("bytes_per_read"))
            e["algorithmic_GBps"] = round(e["algorithmic_bytes"] / (e["k5_ms"] * 1e-3) / 1e9, 1) if e["k5_ms"] else None
            e["frac"] = round(e["algorithmic_GBps"] / HBM_PEAK_GBPS, 4) if e["k5_ms"] else None
            out["entries_path"] = e
        if world == 1 and not a.no_md:
            # The decode a caller with no pileup to follow asks for (bgp_flush_decode(BGP_FETCH_ALL | BGP_DECODE_ONLY): the table and the arenas, nothing
            # for K4): K0 + the decode kernel by their HIP events, against SURVEY 8(d)'s decode bytes.  Beside the headline, not in it.
            for rep in range(5):                                     # (two passes untimed: the legs in front of this one leave the device idle between their kernels)
                if rep == 2:
                    torch.cuda.synchronize(dev); ctx.timing_enable(True); ctx.timing_reset()
                ctx.reset(); ctx.attach_device_stream(sptr, slen)
                ctx.flush(decode=L.BGP_FETCH_ALL | L.BGP_DECODE_ONLY)
            torch.cuda.synchronize(dev)
            tm = ctx.timing_get(); ctx.timing_enable(False)
            k0 = sum(v["total_ms"] for k, v in tm.items() if k.startswith("k0_")) / 3
            dk = {k: v["total_ms"] / 3 for k, v in tm.items() if k.startswith(("k12_", "k1_", "k2_")) and v["launches"]}
            dec_b = int(stream_bytes) + 32 * n_reads + 2 * sum_lseq + 4 * sum_ncig
            kms = sum(dk.values())
            out["decode_only"] = {"what": "bgp_flush_decode(BGP_FETCH_ALL | BGP_DECODE_ONLY): framing + record table + SEQ / QUAL / CIGAR arenas, no plan rows, no side array (3 passes, HIP events)",
                                  "k0_ms": round(k0, 3), "decode_kernels_ms": {k: round(v, 3) for k, v in dk.items()},
                                  "decode_kernel_frac": round(dec_b / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if kms else None,
                                  "decode_8d_frac": round(dec_b / ((k0 + kms) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if kms else None}
            # ... and the pileup a caller asks for who wants no decoded SEQ / QUAL / CIGAR (makePileup over bam.reads: bgp_flush + bgp_pileup_run):
            # K0 + the record table alone + K3 + K4 planning every read from the stream -- no arenas are written.  Beside the headline, never `value`.
            walls = []
            for rep in range(5):
                if rep == 2:
                    torch.cuda.synchronize(dev); ctx.timing_enable(True); ctx.timing_reset()
                t0p = time.perf_counter()
                ctx.reset(); ctx.attach_device_stream(sptr, slen)
                ctx.flush()
                pinfo = ctx.pileup_run()
                torch.cuda.synchronize(dev)
                walls.append((time.perf_counter() - t0p) * 1e3)
            tm = ctx.timing_get(); ctx.timing_enable(False)
            pst = {name: round(sum(v["total_ms"] for k, v in tm.items() if k.startswith(pre) and v["launches"]) / 3, 3) for name, pre in STAGES.items()}
            pst = {k: v for k, v in pst.items() if v}
            out["pileup_only"] = {"what": "bgp_flush + bgp_pileup_run: framing, the record table alone (k1_rows), K3, K4 with every read planned from the stream -- no SEQ / QUAL / CIGAR arenas (3 passes; stages by HIP events, wall by the caller's clock)",
                                  "stages_ms": pst, "kernels_ms": round(sum(pst.values()), 3), "wall_ms": round(sorted(walls[2:])[1], 3), "Mreads_per_s": round(n_reads / (sorted(walls[2:])[1] * 1e-3) / 1e6, 1),
                                  "same_columns_as_the_step": bool(pinfo.n_columns == info.n_columns and pinfo.sum_coverage == info.sum_coverage)}
            ctx.reset(); ctx.attach_device_stream(sptr, slen)       # (the legs below stage their own input)
        if world == 1 and not a.no_e2e and a.config != "cfg5":
            # the same path fed from host memory and from a file -- PCIe- and inflate-inclusive, labelled, never `value`.
            import bench_e2e
            from biod_amd.synth import host_stream_np
            n_e2e = min(a.e2e_reads or 10_000_000, total_reads)
            ctx.reset()
            cfg_e = scaled(a.config, n_e2e)
            st_e = host_stream_np(cfg_e)
            out["e2e"] = {"note": "input NOT resident in HBM: PCIe-inclusive (device_e2e) and inflate-inclusive (file_e2e) rates of the same hot path on %d reads of %s's model; the headline `value` is kernel-only" % (n_e2e, a.config),
                          "device_e2e": bench_e2e.device_e2e(ctx, st_e, n_e2e), "device_e2e_compact_columns": bench_e2e.device_e2e(ctx, st_e, n_e2e, compact=True), "device_e2e_pinned_source": bench_e2e.device_e2e_pinned(ctx, st_e, n_e2e),
                          "file_e2e": bench_e2e.file_e2e(ctx, st_e, n_e2e, int(cfg_e.n_ref)), "file_e2e_streamed": bench_e2e.file_e2e_streamed(ctx, st_e, n_e2e, int(cfg_e.n_ref))}
            del st_e
        if not a.no_cpu_baseline and world == 1:                        # (at N = 1 only: with more ranks the host's cores are the ranks')
            out["cpu_baseline"] = cpu_baseline(a.config, a.cpu_sample_reads)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), file=result_out, flush=True)
        if ("e2e" in out and not out["e2e"]["file_e2e_streamed"]["equals_one_shot"]["equal"]) or (checks is not None and not checks["ok"]) or hist_ok is False or ("parity_interior" in out and not out["parity_interior"]["all_equal"]) or ("md_path" in out and not (out["md_path"]["prefix_vs_oracle"]["equal"] and out["md_path"].get("whole_stream_vs_generator_reference", {"equal": True})["equal"])):
            sys.stderr.write("bench.py: a result check FAILED (see checks_full_size / hist_reduce_equals_sum_of_ranks / parity_interior)\n")
            ctx.close()
            sys.exit(3)
    ctx.close()
    if with_comm:
        if comm is not None:
            rccl.ncclCommDestroy(comm)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
