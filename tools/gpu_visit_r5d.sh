cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pe.py tests/test_gpu_infer.py -q -rP 2>&1 | grep -E "passed|failed|^pe |^configs|Error|assert" | cut -c1-330 | tee gpurun_out/r5d_pe_tests.txt
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -rP -k "denominator or pe" 2>&1 | grep -E "passed|failed|^bench|Error|assert" | cut -c1-330 | tee -a gpurun_out/r5d_pe_tests.txt
timeout 900 python bench.py > gpurun_out/r5d_bench.json 2> gpurun_out/r5d_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r5d_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5d_bench.json'))
print('value',d['value'],'batched',d['batched']['value'],'plms',d['plms_50']['value'],'train',d['train_step'].get('ms_per_step'),'var',d['train_step'].get('variable_shape_ms'))
print('stages',{k:v for k,v in d['stages'].items() if k.endswith('_ms')})
print('res_skip',d.get('roofline_res_skip'))
print('fit',d['batched'].get('mel_error_vs_reference'), d['batched'].get('mel_error_vs_reference_missing'))
PY
