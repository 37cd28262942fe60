#!/bin/bash
# round 5 batch 8: grouped K/V at head dim 256 on the two-wave backward kernels (parity + time against the feature-half kernels), XLNet surgery
# on the vector path, the limiter probe with its full output
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5h}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_surgery.py tests/test_gpu_parity.py -m gpu -x -q -k "surgery or xlnet or grouped or head_dim_256 or d256 or gqa" 2>&1 | tail -4 | tee $O/pytest.log
python tools/bench_gqa_train.py 2>&1 | tail -8 | tee $O/gqa_train.log
python - <<'PY' 2>&1 | tee $O/gqa_d256.log
import torch, time
import flash_attention_softmax_n_amd as pkg
dev = torch.device("cuda:0")
for (B, H, Hkv, S) in ((4, 16, 4, 4096), (2, 32, 4, 4096)):
    D = 256
    q = torch.randn(B, H, S, D, device=dev, dtype=torch.bfloat16).mul_(0.5).requires_grad_()
    k, v = (torch.randn(B, Hkv, S, D, device=dev, dtype=torch.bfloat16).mul_(0.5).requires_grad_() for _ in range(2))
    do = torch.randn(B, H, S, D, device=dev, dtype=torch.bfloat16)
    for causal in (False, True):
        def step():
            q.grad = k.grad = v.grad = None
            pkg.flash_attention_n(q, k, v, softmax_n_param=1.0, is_causal=causal).backward(do)
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        fl = 3.5 * 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        print(f"GQA D=256 (B={B},H={H},Hkv={Hkv},S={S}) causal={causal}: fwd+bwd {dt*1e3:.3f} ms  {fl/dt/1e12:.0f} TFLOP/s algorithmic")
PY
cd /tmp && export TMPDIR=/tmp LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
{
echo "=== MFMA-only launch"; bash $R/tools/limiter_probe.sh mfma
echo "=== M0 forward loop"; bash $R/tools/limiter_probe.sh 8 16 4096 4096 64 1 0 0 12000
} > $O/limiter.log 2>&1
