"""The batched-staging state machine under random call sequences: pushes, pushes from page-locked memory and acquire / commit pairs of random sizes (0-byte commits, commits
that fill a buffer exactly, buffers left almost empty), windows taken none, one or all at a time, random batch sizes -- the columns handed out
== the columns of the stream staged at once.  usage: python scratch/fuzz_batch_api.py [rounds] [seed0]"""
import ctypes as C, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import test_gpu_parity as T
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import host_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
ctx = Context(0); lib, h = L.lib(), ctx.handle
hip = C.CDLL("libamdhip64.so")
PIN_CAP = 8 << 20
pin = C.c_void_p(); assert hip.hipHostMalloc(C.byref(pin), C.c_size_t(PIN_CAP), C.c_uint(0)) == 0      # (the stream's page-locked twin)
npin = 0
bad = 0; ran = 0; nbat = 0; t0 = time.time()
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    cfg = T.make_cfg(n_reads=int(rng.integers(300, 4000)), n_ref=int(rng.integers(1, 4)), ref_len=int(rng.integers(20_000, 60_000)), read_len=int(rng.integers(50, 151)), profile=int(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)))
    try:
        st = host_stream(cfg)
    except L.BiodGpuError:
        continue                                   # (a shape the generator refuses)
    T.stage(ctx, st)
    info = ctx.pileup_run()
    want = T.device_columns(ctx, info)
    batch = int(rng.choice([4096, int(rng.integers(4096, 70000)), int(rng.integers(70000, 400000))]))
    got_pos, got_cols = [], []
    def take(limit):
        cw = L.Columns()
        while limit:
            limit -= 1
            rc = L.check(lib.bgp_batch_next_window(h, C.byref(cw)), h)
            if rc != L.BGP_OK:
                return rc
            cols = L.as_array(cw.columns, cw.n_positions, L.COLUMN_DTYPE)
            got_cols.append(cols.copy())
        return L.BGP_OK
    try:
        L.check(lib.bgp_batch_begin(h, batch, 0, int(rng.integers(100, 5000))), h)
        off = 0; buf = np.frombuffer(st, np.uint8)
        use_pin = len(st) <= PIN_CAP and rng.random() < 0.6
        if use_pin: C.memmove(pin.value, buf.ctypes.data, len(st))
        while off < len(st):
            u = rng.random()
            if use_pin and u < 0.4:                  # from where the bytes lie: all of them are taken, or none (EAGAIN: not right behind what waits -- cannot happen here -- or a caller's buffer is out)
                m = int(min(len(st) - off, rng.choice([1, int(rng.integers(1, 3 * batch)), len(st) - off])))
                taken = C.c_size_t()
                rc = L.check(lib.bgp_batch_push_pinned(h, C.c_void_p(pin.value + off), m, C.byref(taken)), h)
                assert taken.value in (0, m), (taken.value, m)
                off += taken.value; npin += taken.value != 0
                if rc == L.BGP_EAGAIN: take(int(rng.integers(1, 4)))
            elif u < 0.7:
                m = int(min(len(st) - off, rng.integers(1, 3 * batch)))
                taken = C.c_size_t()
                rc = L.check(lib.bgp_batch_push(h, C.c_void_p(buf.ctypes.data + off), m, C.byref(taken)), h)
                off += taken.value
                if rc == L.BGP_EAGAIN: take(int(rng.integers(1, 4)))
            else:
                ptr, cap = C.c_void_p(), C.c_size_t()
                rc = L.check(lib.bgp_batch_acquire(h, C.byref(ptr), C.byref(cap)), h)
                if rc == L.BGP_EAGAIN:
                    take(int(rng.integers(1, 4))); continue
                m = int(min(cap.value, len(st) - off, rng.choice([0, 1, int(rng.integers(0, cap.value + 1)), cap.value])))
                C.memmove(ptr.value, buf.ctypes.data + off, m)
                if rng.random() < 0.3: take(1)                       # windows may be taken while the buffer is the caller's
                L.check(lib.bgp_batch_commit(h, m), h)
                off += m
            if rng.random() < 0.4: take(int(rng.integers(0, 3)))
        L.check(lib.bgp_batch_end(h), h)
        rc = take(1 << 30)
        assert rc == L.BGP_EOF, rc
        ran += 1; nbat += -(-len(st) // batch)
        got = np.concatenate(got_cols) if got_cols else np.zeros(0, L.COLUMN_DTYPE)
        assert len(got) == len(want), (len(got), len(want))
        for f in L.COLUMN_DTYPE.names:
            assert np.array_equal(got[f], want[f]), f
    except Exception as e:
        bad += 1; print("BATCHAPI seed", seed0 + k, "batch", batch, "len", len(st), type(e).__name__, str(e)[:200])
        ctx.reset()
print(n, "rounds,", ran, "ran,", nbat, "batches (at least),", npin, "page-locked pushes,", bad, "failures,", round(time.time() - t0, 1), "s")
