#!/bin/bash
cd /root/repo
cp biod_amd/csrc/frame.hip /tmp/frame.hip.orig
python - <<'PY'
p='/root/repo/biod_amd/csrc/frame.hip'
s=open(p).read()
s=s.replace("        ++cnt; ls += (uint64_t)lseq; nc += n_cig;\n        long_seen","        if (cnt < 128) a.slots[k * 128 + cnt] = (uint32_t)(off - k * (uint64_t)a.seg_bytes);\n        ++cnt; ls += (uint64_t)lseq; nc += n_cig;\n        long_seen")
s=s.replace("    uint32_t* status; /* [0] status bits, [1] dirty counter */\n};","    uint32_t* status; /* [0] status bits, [1] dirty counter */\n    uint32_t* slots;\n};")
s=s.replace("    a.status = c->d_status.as<uint32_t>();\n","    a.status = c->d_status.as<uint32_t>();\n    BGP_TRY(c->d_seg_slots.reserve(n_seg * 128 * 4)); a.slots = c->d_seg_slots.as<uint32_t>();\n",1)
open(p,'w').write(s)
p='/root/repo/biod_amd/csrc/common.h'
s=open(p).read()
s=s.replace("    DevBuf d_seg_rec_base, d_seg_seq_base, d_seg_cig_base;\n","    DevBuf d_seg_rec_base, d_seg_seq_base, d_seg_cig_base, d_seg_slots;\n")
open(p,'w').write(s)
PY
cp biod_amd/csrc/common.h /tmp/common.h.new
make -C biod_amd/csrc -j8 2>&1 | grep -E "error" -A3
python bench.py --reads 50000000 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('with slot store', d['ms_per_step'], d['kernels']['k0_chase'], d['kernels']['k1_emit'])"
cp /tmp/frame.hip.orig biod_amd/csrc/frame.hip
git checkout biod_amd/csrc/common.h 2>/dev/null
