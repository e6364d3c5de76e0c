import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle_lib as O
from biod_amd import _lib as L, bamio
from biod_amd.pileup import Context
from test_gpu_parity import dense_from_oracle, device_columns
ctx=Context(0)
hdr, st = bamio.read_bam(os.path.join(os.path.dirname(os.path.abspath(__file__)),'..','tests','golden','data','ion_20_chunk.bam'))
ctx.reset(); ctx.push_inflated(st); ctx.flush()
info=ctx.pileup_run()
ora=O.pileup(st)
dense=device_columns(ctx, info); exp=dense_from_oracle(ora, info.runs)
bad=np.nonzero(dense['coverage']!=exp['coverage'])[0]
print(len(dense), len(bad), bad[:20])
soa=O.decode_soa(st)
print(list(zip(soa.pos[:20], soa.end_pos[:20], soa.flag_nc[:20]&0xFFFF, soa.l_seq[:20])))
for b in bad[:5]: print(b, dense[b], exp[b])
