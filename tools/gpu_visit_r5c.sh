cd $GRAFT_REPO_ROOT
timeout 600 python tools/gpu_tail_tiling.py f16_w6 32 2>&1 | grep tail_tiling | tee gpurun_out/r5c_tail_tiling.txt
timeout 300 python tools/gpu_nt_diag.py 2>&1 | grep -E "^f16_w6 \{\}|differ" | tee gpurun_out/r5c_nt_diag.txt
timeout 600 python -m pytest tests/test_gpu_headline.py -q -k "mid_size" -rP 2>&1 | grep -E "passed|failed|^fused" | cut -c1-300
