cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_embed.py tests/test_gpu_cdriver.py -q -x 2>&1 | tail -6 > gpurun_out/r04_t3.log
for w in 0 1 0 1; do VD_BWD0_WIDE=$w python bench.py --no-cpu-baseline --no-extra-legs --sustain-seconds 0 --eval-epochs 0 --steps 30 --no-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DM wide=$w', d['value'], d['ms_per_step'], d['ms_per_step_median'])" >> gpurun_out/r04_ab_bwd0.txt; done
for w in 0 1; do VD_BWD0_WIDE=$w python bench.py --method mtt --classes 400 --frames 8 --size 64 --steps 5 --warmup 2 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MTT wide=$w', d['value'], d['ms_per_step'])" >> gpurun_out/r04_ab_bwd0.txt; done
for w in 0 1; do VD_BWD0_WIDE=$w python tools/syn_side_trace.py 30 > /dev/null; VD_BWD0_WIDE=$w python - >> gpurun_out/r04_ab_bwd0.txt <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from video_distillation_amd import distill, plan
from video_distillation_amd.networks import _batch_hint
dev = torch.device("cuda:0"); geo = plan.NetGeometry(16, 112, 112); C = 50
be = distill.HipBackend(geo, dev, chunk=3200, syn_batch_hint=_batch_hint(C))
syn = torch.randn(C, 16, 3, 112, 112, device=dev); f_real = torch.randn(C, geo.num_feat, device=dev)
w = be.new_network(seed=0); be._dither = 8
f_syn, handle = be.embed_syn(syn, w); loss_c, g_syn = be.dm_loss(f_real, f_syn, C)
for _ in range(3): be.embed_backward(handle, g_syn)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): be.embed_backward(handle, g_syn)
torch.cuda.synchronize(); print("embed_backward 50 clips wide=%s: %.3f ms" % (os.environ.get("VD_BWD0_WIDE"), (time.perf_counter() - t0) / 20 * 1e3))
PY
done
cat gpurun_out/r04_ab_bwd0.txt
