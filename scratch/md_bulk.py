"""Bulk (fetch-to-fetch) form of the MD provider chain, validated against the oracle.
Between two chunk fetches nothing depends on the order of reads except 'first argmax of end among the
reads that pass add()'s filter' -- so the controller can jump from event to event."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/scratch')
import numpy as np
import oracle_lib as O
from md_proto import oracle_reads, random_stream, INF


def bulk(reads, skip_zero):
    n = len(reads)
    if n == 0:
        return {}
    ref = np.array([r['ref'] for r in reads]); pos = np.array([r['pos'] for r in reads], dtype=np.int64); end = np.array([r['end'] for r in reads], dtype=np.int64)
    L_of = [len(r['dna']) for r in reads]
    intervals = []; visited = []
    # persistent machine state
    P = -1; L = 0; ci = 0; CE = 0; has_next = False; NX = -1; curr_ref = None
    i = 0
    while i < n:
        r0 = i
        rref = ref[i]
        r1 = r0
        while r1 < n and ref[r1] == rref:
            r1 += 1
        # per-run precomputation (parallel on the device): group starts, max end before each group, had_zero
        gstart = np.ones(r1 - r0, bool); gstart[1:] = pos[r0 + 1:r1] != pos[r0:r1 - 1]
        pm = np.maximum.accumulate(end[r0:r1])                    # inclusive prefix max of end
        mb = np.zeros(r1 - r0, np.int64)                          # max end over reads of EARLIER groups
        last = 0
        for k in range(r1 - r0):
            if gstart[k]:
                last = pm[k - 1] if k > 0 else 0
            mb[k] = last
        hz = mb <= pos[r0:r1] - 1                                  # coverage(pos-1) == 0  (first group: True)
        run_end = int(pm[-1])

        def lidx(j):
            return j - r0

        def admit(j_from, j_to, force_hz):
            """add() for reads [j_from, j_to) under the current CE; returns nothing, updates has_next/NX/curr_ref"""
            nonlocal has_next, NX, curr_ref
            for j in range(j_from, j_to):                          # (device: masked wave argmax; here literal but order-free)
                if ref[j] != curr_ref:
                    curr_ref = ref[j]; has_next = True; NX = j; continue
                h = True if force_hz else bool(hz[lidx(j)])
                if pos[j] > CE and not h:
                    continue
                if end[j] > CE:
                    if not has_next:
                        has_next = True; NX = j
                    elif end[j] > end[NX]:
                        NX = j

        def group_end(j):
            k = j
            while k < r1 and pos[k] == pos[j]:
                k += 1
            return k

        def init_at(j):
            """initNewReference at the group starting at read j"""
            nonlocal i, P, L, ci, CE, has_next
            x = int(pos[j]); ge = group_end(j)
            admit(j, ge, True); i = ge
            if has_next:
                P = NX; L = L_of[P]; ci = 0; CE = int(end[P]); has_next = False
                if L == 0:
                    return None
                intervals.append([rref, x, P, 0]); ci = 1
            else:
                intervals.append([rref, x, -1, 0])
                if P >= 0 and ci < L:
                    intervals.append([rref, x + 1, P, ci])
            return x

        y = init_at(i)
        if y is None:
            return 'ERR'
        seg_from = y
        while True:
            # ---- find the next event under the current CE
            # candidates among not yet admitted reads of the run
            jc = -1
            if not has_next:
                for j in range(i, r1):
                    if (pos[j] <= CE or hz[lidx(j)]) and end[j] > CE:
                        jc = j; break
                    if skip_zero and hz[lidx(j)] and gstart[lidx(j)] and j > i - 0 and False:
                        pass
            # next gap group (skip_zero only): first group start j >= i with had_zero
            jg = -1
            if skip_zero:
                for j in range(i, r1):
                    if gstart[lidx(j)] and hz[lidx(j)]:
                        jg = j; break
            empty_at = y + max(1, L - ci)                          # first visited column whose check sees an empty chunk
            if has_next:
                F = empty_at
            elif jc >= 0:
                F = max(empty_at, int(pos[jc]))
            else:
                F = INF
            # where does coverage end?  gap before group jg at E = mb[jg]; run end at run_end
            if jg >= 0:
                E = max(int(mb[lidx(jg)]), y + 1)
                # a candidate found AFTER the gap group does not count before the jump: restrict
                if jc >= 0 and jc >= jg and not has_next:
                    F = INF if True else F
            else:
                E = max(run_end, y + 1) if True else INF
            # admissions before the event
            if F < E:
                # fetch at column F: admit every group with pos <= F
                j = i
                while j < r1 and pos[j] <= F and not (skip_zero and j == jg):
                    j += 1
                admit(i, j, False); i = j
                ci = min(L, ci + (F - y))
                assert has_next
                P = NX; L = L_of[P]; CE = int(end[P]); has_next = False
                ci = min(F - int(pos[P]), L)
                if ci >= L:
                    return 'ERR'
                intervals.append([rref, F, P, ci]); ci += 1
                y = F
                continue
            # coverage ends at E first (or no fetch at all)
            if jg >= 0:
                # admit everything before the gap group, then jump
                admit(i, jg, False); i = jg
                ci = min(L, ci + (E - y))
                visited.append((rref, seg_from, E - 1))
                yy = init_at(jg)
                if yy is None:
                    return 'ERR'
                y = yy; seg_from = y
                continue
            # run end
            admit(i, r1, False); i = r1
            ci = min(L, ci + (E - y))
            visited.append((rref, seg_from, E - 1))
            break
    res = {}
    for rf, a, b in visited:
        for col in range(a, b + 1):
            res[(rf, col)] = 'N'
    for k, (rf, col, prov, c0) in enumerate(intervals):
        endc = INF
        if k + 1 < len(intervals) and intervals[k + 1][0] == rf:
            endc = intervals[k + 1][1]
        if prov < 0:
            continue
        d = reads[prov]['dna']; c = col; j = c0
        while c < endc and j < len(d) and (rf, c) in res:
            res[(rf, c)] = d[j]; j += 1; c += 1
    return res


def compare(st, skip_zero):
    try:
        ora = O.pileup(st, O.MODE_PILEUP_COLUMNS, use_md=True, skip_zero=skip_zero)
        exp = {(int(r), int(p)): chr(b) for r, p, b in zip(ora.ref, ora.pos, ora.refbase)}
    except O.OracleError as e:
        exp = 'ERR' if e.code == O.ORC_EMD else ('OTHER', e.code)
    got = bulk(oracle_reads(st), skip_zero)
    if exp == 'ERR' or got == 'ERR':
        return exp == got, (exp if isinstance(exp, str) else len(exp), got if isinstance(got, str) else len(got))
    if isinstance(exp, tuple):
        return True, exp
    bad = [(k, exp[k], got.get(k)) for k in sorted(exp) if got.get(k) != exp[k]]
    extra = [k for k in got if k not in exp]
    return (not bad and not extra), (bad[:5], extra[:5], len(exp))


if __name__ == '__main__':
    rng = np.random.default_rng(11)
    total = fails = 0
    cfgs = [dict(), dict(p_gap=0.1), dict(p_n=0.1), dict(p_nomd=0.05), dict(p_badmd=0.05), dict(p_gap=0.05, p_n=0.05, p_nomd=0.02, p_badmd=0.02, nref=3)]
    for trial in range(400):
        kw = cfgs[trial % len(cfgs)]
        st = random_stream(rng, int(rng.integers(5, 120)), **kw)
        for skip in (True, False):
            ok, info = compare(st, skip)
            total += 1
            if not ok:
                fails += 1
                if fails <= 6:
                    print('MISMATCH trial', trial, kw, 'skip', skip, info)
    print('total', total, 'fails', fails)
