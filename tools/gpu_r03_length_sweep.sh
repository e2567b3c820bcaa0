mkdir -p gpurun_out; : > gpurun_out/r03_length_sweep.txt
for s in 4.0 3.5 4.5 3.0 5.0; do
  timeout 120 python bench.py --seconds $s --steps 2 --warmup 1 --no-others --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; k=d['kernel_classes_one_eval']['conv3x3_wide']
print('seconds $s', 'T', c['T'], 'tiles_x', c['T']//32, 'utt/s', round(d['value'],4), 'ms/step', round(d['ms_per_step'],1), 'us per frame per utterance-batch', round(d['ms_per_step']*1e3/c['T'],2), 'dominant class ms', k['ms'], 'TFLOP/s', round(k['tflops'],1))" | tee -a gpurun_out/r03_length_sweep.txt
done
