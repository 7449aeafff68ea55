out=gpurun_out/r04; mkdir -p $out
CS_BENCH_SHARD_DEVICES=0,0 python bench.py --gpus 2 --steps 50 --warmup 5 > $out/bench_2shards_1gpu.json 2> $out/bench_2shards.err || { tail -30 $out/bench_2shards.err; cat $out/bench_2shards_1gpu.json | cut -c1-3000; exit 1; }
python3 -c "
import json; d=json.load(open('$out/bench_2shards_1gpu.json'))
print('N=2 rehearsal value', d['value'], 'ms', d['ms_per_step']); print(json.dumps(d['multi_gpu_checks'])); print(json.dumps(d['rccl'])); print(json.dumps(d.get('default_routing'))); print(json.dumps(d.get('config_5')))
"
CS_REAL_MODEL_DIR=build/synth_model python -m pytest tests/test_gpu_real_model.py -q -m gpu -k "not semantic" 2>&1 | tail -5 | tee $out/real_model_rehearsal.log
