#!/bin/bash
# regression sweep of the final build: soak (loss falls, memory flat), AViNet, config 5, inference, harness
cd /root/repo
python tools/soak.py 60 8 2>&1 | grep -v amdgpu | tail -4
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from vinet_amd import engine, loss as VL, model as VM, optim as VO, synth
engine.set_default_dtype("bf16")
dev = torch.device("cuda:0")
m = VM.VideoSaliencyModel(num_clips=32); m.load_state_dict(synth.synth_state_dict(m.state_dict(), 0)); m = m.to(dev).train()
x = synth.clip(32, 32, 224, 384, 1).to(dev).permute(0, 2, 1, 3, 4); gt = synth.gt_map(32, 224, 384, 1).to(dev)
opt = VO.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
res = []
for i in range(24):
    opt.zero_grad(); l = VL.kldiv(m(x), gt); l.backward(); opt.step()
    if i in (3, 23):
        torch.cuda.synchronize(); res.append((torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))
print("memory after step 4 / 24 (allocated, reserved GiB):", res)
assert abs(res[0][0] - res[1][0]) < 0.05 and abs(res[0][1] - res[1][1]) < 1.0, "memory grows across steps"
PY
python bench.py --model avinet --no-sweep --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('avinet', d['value'], d['ms_per_step'])"
python bench.py --clip 64 --height 256 --width 448 --no-sweep --no-cpu-baseline --steps 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5', d['value'], d['ms_per_step'], d['roofline']['whole_step'])"
python bench.py --mode infer --batch 64 --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer b64', d['value'])"
python bench.py --mode infer --batch 1 --graph --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer b1 graph', d['value'])"
python -m vinet_amd.generate_result --synthetic_frames 191 --allow_synthetic_weights 2>&1 | grep -i "fps\|frames" | tail -2
