"""Config 4 in its FILE form: ONE coordinate-sorted, indexed BAM on disk, N ranks (a rank a GPU), every rank cuts its share by the .bai
(shard.plan_file_shards: ReferenceSequence.startVirtualOffset, reference.d:96-104; Index.getMinimumOffset, baifile.d:77-82), loads its
regions with the blocks inflated on ITS GPU (bgp_load_bam_region_device: bam[ref][beg .. end], randomaccessmanager.d:222-244,366-462),
piles them up (makePileup(reads, false, beg, end): pileup.d:683-694) and the whole-BAM depth histogram is summed over the ranks
(bgp_hist_reduce, RCCL over xGMI -- or, REHEARSAL on a box with fewer GPUs than ranks, by this script over gloo).  No data-path
collective.  Prints ONE JSON line on rank 0: whole-job Mreads/s by the slowest rank's clock (barrier + synchronize on both sides).
Run:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_cfg4_file.py --reads R [--path F]
      (python bench_cfg4_file.py --gpus N ... spawns the launcher itself).  The file is written (and indexed on the device) by rank 0."""
import argparse, ctypes as C, json, os, socket, subprocess, sys, time
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1); ap.add_argument("--reads", type=int, default=8_000_000); ap.add_argument("--path", default="/tmp/cfg4.bam")
    ap.add_argument("--reps", type=int, default=3); ap.add_argument("--host-inflate", action="store_true", help="bgp_load_bam_region (zlib on the host's threads) instead of the device inflate")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    out = os.fdopen(os.dup(1), "w"); os.dup2(2, 1)
    import numpy as np, torch, torch.distributed as dist
    from biod_amd import _lib as L, bamio, shard
    from biod_amd.checks import DevArray
    from biod_amd.pileup import BaiFile, Context
    from biod_amd.synth import host_stream_np, scaled
    ngpu = torch.cuda.device_count()
    rehearsal = world > ngpu
    dev = torch.device("cuda", local % max(ngpu, 1)); torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo" if (rehearsal or world == 1) else "nccl", rank=rank, world_size=world, **({} if (rehearsal or world == 1) else {"device_id": dev}))
    comm = rccl = None
    if world > 1 and not rehearsal:
        sys.path.insert(0, ROOT); import bench
        rccl, comm = bench.rccl_communicator(torch, dist, rank, world, dev)
    cfg = scaled("cfg3", a.reads)
    ctx = Context(dev.index)
    if rank == 0:                                         # the file and its index (createIndex on the device), once
        from concurrent.futures import ThreadPoolExecutor
        SL = 2_000_000
        with open(a.path, "wb") as f, ThreadPoolExecutor(16) as ex:
            f.write(bamio.bgzf_block(bamio.bam_header_bytes([(f"chr{i + 1}", int(cfg.ref_len)) for i in range(max(1, cfg.n_ref))])))
            carry = b""
            for first in range(0, a.reads, SL):
                st = carry + host_stream_np(cfg, first, min(SL, a.reads - first)).tobytes()
                whole = len(st) - len(st) % 64000 if first + SL < a.reads else len(st)
                b = memoryview(st)
                for blk in ex.map(lambda at: bamio.bgzf_block(bytes(b[at:min(at + 64000, whole)])), range(0, whole, 64000)):
                    f.write(blk)
                carry = st[whole:]
            f.write(bamio.BGZF_EOF)
        ctx.load_bam(a.path, 0); ctx.flush()
        open(a.path + ".bai", "wb").write(ctx.bai_build())
        whole_info = ctx.pileup_run()
        whole = (int(ctx.num_records()), int(whole_info.n_columns), int(whole_info.sum_coverage))
        whole_hist = torch.as_tensor(DevArray(whole_info.device_depth_hist, (L.BGP_DEPTH_BINS,), "<i8"), device=dev).clone().cpu()
        ctx.reset()
    dist.barrier()
    bai = BaiFile(a.path)
    lens = [int(cfg.ref_len)] * max(1, int(cfg.n_ref))
    mine = shard.plan_file_shards(bai, lens, world)[rank]
    lib = L.lib()

    def step():
        """this rank's regions: load (device inflate), frame + table, pile up; columns of [beg, end) counted from the device's dense array"""
        hist = torch.zeros(L.BGP_DEPTH_BINS, dtype=torch.int64, device=dev)
        n_rec = n_col = s_cov = 0
        for ref_id, beg, end in mine:
            ctx.reset()
            (ctx.load_bam_region if a.host_inflate else ctx.load_bam_region_device)(*((a.path, bai, ref_id, beg, end, 0) if a.host_inflate else (a.path, bai, ref_id, beg, end)))
            ctx.flush()
            if ctx.num_records() == 0:
                continue
            info = ctx.pileup_run()
            cols = torch.as_tensor(DevArray(info.device_columns, (int(info.n_positions), 8), "<u4"), device=dev)
            for r in info.runs:
                if int(r["ref_id"]) != ref_id:
                    continue
                lo, hi = max(int(r["start_pos"]), beg), min(int(r["end_pos"]), end)
                if lo < hi:
                    cov = cols[int(r["column_base"]) + lo - int(r["start_pos"]): int(r["column_base"]) + hi - int(r["start_pos"]), 0].to(torch.int64)
                    hist += torch.bincount(cov.clamp(max=L.BGP_DEPTH_BINS - 1), minlength=L.BGP_DEPTH_BINS)
                    n_col += int((cov > 0).sum().item()); s_cov += int(cov.sum().item())
            # reads that START in [beg, end): every read of the file is counted once over the ranks
            rec = torch.as_tensor(DevArray(ctx.decode_device(0).rec, (int(ctx.num_records()), 8), "<i4"), device=dev)
            n_rec += int(((rec[:, 0] == ref_id) & (rec[:, 1] >= beg) & (rec[:, 1] < end)).sum().item())
        if world > 1 and not rehearsal:
            L.check(lib.bgp_hist_reduce(ctx.handle, comm, C.c_void_p(hist.data_ptr()), L.BGP_DEPTH_BINS), ctx.handle)
        elif world > 1:
            h = hist.cpu(); dist.all_reduce(h); hist.copy_(h)
        return hist, n_rec, n_col, s_cov

    times = []
    for rep in range(a.reps + 1):
        torch.cuda.synchronize(dev); dist.barrier()
        t0 = time.perf_counter()
        hist, n_rec, n_col, s_cov = step()
        torch.cuda.synchronize(dev); dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64); dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        if rep:
            times.append(float(dt.item()))
    tot = torch.tensor([n_rec, n_col, s_cov], dtype=torch.int64); dist.all_reduce(tot)
    if rank == 0:
        best = min(times)
        unmapped_or_uncounted = whole[0] - int(tot[0])
        ok = int(tot[1]) == whole[1] and int(tot[2]) == whole[2] and bool(torch.equal(hist.cpu()[1:], whole_hist[1:])) and unmapped_or_uncounted == 0
        print(json.dumps({"what": "config 4, file form: one indexed BAM, ranks cut by the .bai, regions loaded with " + ("zlib on the host's threads" if a.host_inflate else "the blocks inflated on each rank's GPU") + ", depth histogram summed over the ranks",
                          "n_gpus": world, "gpus_on_the_box": ngpu, "REHEARSAL": bool(rehearsal), "reads": a.reads, "file_bytes": os.path.getsize(a.path), "regions_of_rank_0": len(mine),
                          "Mreads_per_s": round(a.reads / best / 1e6, 2), "best_s": round(best, 4), "runs_s": [round(t, 4) for t in times], "scaling": "strong",
                          "equals_whole_file": ok, "reads_counted": int(tot[0]), "covered_columns": int(tot[1]), "sum_coverage": int(tot[2])}), file=out, flush=True)
        for p in (a.path, a.path + ".bai"):
            try: os.unlink(p)
            except OSError: pass
        if not ok:
            sys.exit(3)
    ctx.close()
    if comm is not None:
        rccl.ncclCommDestroy(comm)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
