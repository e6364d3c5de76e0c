"""Size-independent properties of a finished pileup, evaluated on the device (torch views of the ctx's HBM
arrays, no copies): what bench.py reports as `checks_full_size` and what the full-size `-m gpu` tests assert.
No oracle in here -- these are identities of the domain: every covering read starts exactly once, every read
contributes basesCovered() entries (pileup.d:352: it is in the columns position .. end_position - 1), a column's
coverage is the sum of its base classes (pileup.d:115-122: every read shows a base or '-')."""
from __future__ import annotations


class DevArray:
    """A device pointer as something torch.as_tensor understands (zero copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape), "typestr": typestr, "version": 2}


def device_view(torch, dev, ptr, shape, typestr):
    return torch.as_tensor(DevArray(ptr, shape, typestr), device=dev)


def full_size_invariants(torch, dev, ctx, info, n_reads=None, own_histogram=True):
    """own_histogram=False: the depth histogram on the device is no longer this rank's own (it has been summed over the ranks in place:
    bench.py then holds it against the job's totals instead)."""
    r = ctx.decode_device(0)
    rec = device_view(torch, dev, r.rec, (int(r.n), 8), "<i4")       # ref_id,pos,end_pos,bin_mq_nl,flag_nc,l_seq,rec_off(2)
    covered = (rec[:, 2] - rec[:, 1]).to(torch.int64)
    reads_used, bases_covered = int((covered != 0).sum().item()), int(covered.sum().item())
    key = (rec[:, 0].to(torch.int64) << 32) | rec[:, 1].to(torch.int64)
    is_sorted = bool((key[1:] >= key[:-1]).all().item()) if int(r.n) > 1 else True
    cols = device_view(torch, dev, info.device_columns, (int(info.n_positions), 8), "<i4") if info.n_positions else None
    s_cov = s_nst = n_col = mx = 0
    cls_ok = True
    step = 1 << 25
    for o in range(0, int(info.n_positions), step):
        c = cols[o:o + step]
        cov = c[:, 0].to(torch.int64)
        s_cov += int(cov.sum().item()); s_nst += int(c[:, 1].to(torch.int64).sum().item()); n_col += int((cov > 0).sum().item())
        mx = max(mx, int(cov.max().item()))
        cls_ok = cls_ok and bool((c[:, 2:8].to(torch.int64).sum(dim=1) == cov).all().item())
    hist = device_view(torch, dev, info.device_depth_hist, (65536,), "<i8")
    hist_ok = (int(hist.sum().item()) == int(info.n_positions) and int((hist * torch.arange(65536, device=dev)).sum().item()) == s_cov if mx < 65535 else True) if own_histogram else True
    ok = (s_nst == reads_used == info.n_reads_used and s_cov == bases_covered == info.sum_coverage and n_col == info.n_columns
          and mx == info.max_coverage and cls_ok and is_sorted and hist_ok and (n_reads is None or int(r.n) == n_reads))
    return {"reads_used": reads_used, "sum_n_starting_here": s_nst, "sum_basesCovered": bases_covered, "sum_coverage": s_cov,
            "ok": bool(ok), "coverage_equals_class_sum_everywhere": cls_ok, "depth_histogram_consistent": bool(hist_ok)}
