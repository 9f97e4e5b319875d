mkdir -p gpurun_out/final2
python bench.py 2> gpurun_out/final2/bench_default.err | grep "^{" > gpurun_out/final2/bench_default.json
python -c "
import json; d=json.load(open('gpurun_out/final2/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['training_config3_one_gpu']['images_per_s'])"
python tools/train_breakdown.py --detail conv_wgrad_kernel,selscan_bwd_chunk,dwconv3x3_wgrad > gpurun_out/final2/train_breakdown.txt 2>&1; grep "wall\|GPU kernel" gpurun_out/final2/train_breakdown.txt
python tools/bench_conv_wgrad.py > gpurun_out/final2/bench_conv_wgrad.txt 2>/dev/null; cat gpurun_out/final2/bench_conv_wgrad.txt
