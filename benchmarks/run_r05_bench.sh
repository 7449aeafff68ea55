#!/bin/bash
# Round 5: the bench lines as the driver issues them (N = 1), and the N-shard one-process form rehearsed on one GPU
# (CS_BENCH_SHARD_DEVICES=0,0: two shards on device 0; the RCCL child runs at world 1 with the exchange forced).
set -o pipefail
out=gpurun_out/r05
mkdir -p $out
python bench.py > $out/bench_n1.json 2> $out/bench_n1.err || { tail -30 $out/bench_n1.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/bench_n1.json'))
e=d['encoder']; es=d['embed_search']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'])
print('default_routing', json.dumps(d.get('default_routing',{}).get('per_k')))
print('encoder mean ms', e['ms_per_batch'], 'cls ms', e['cls_pool_variant']['ms_per_batch'], 'l512 ms', e['encoder_l512']['ms_per_batch'])
print('l512 per kernel', e['encoder_l512']['per_kernel_us_per_layer'])
print('mean per kernel', e['roofline']['per_kernel_us_per_layer'])
print('embed_search', {k: v for k, v in es.items() if not isinstance(v, (dict, str))})
print('cls embed_search', es['cls_pool_variant'])
print('config_1m', d.get('config_1m')); print('e2e', d.get('e2e_index_search'))
"
CS_BENCH_SHARD_DEVICES=0,0 python bench.py --gpus 2 --steps 50 --warmup 5 > $out/bench_2shards_1gpu.json 2> $out/bench_2shards.err || { tail -30 $out/bench_2shards.err; cat $out/bench_2shards_1gpu.json | cut -c1-3000; exit 1; }
python3 -c "
import json; d=json.loads(open('$out/bench_2shards_1gpu.json').read().strip().split('\n')[-1])  # (the record is printed before the optional legs and again, complete, behind them)
print('N=2 rehearsal value', d['value'], 'ms', d['ms_per_step']); print(json.dumps(d['multi_gpu_checks'])); print(json.dumps(d['rccl'])); print(json.dumps(d.get('default_routing'))); print(json.dumps(d.get('config_5')))
"
# the operator-runnable real-model check, rehearsed on a stand-in directory (seeded random weights written as a HF
# snapshot by tests/golden/make_synthetic_model_dir.py; golden.npz from transformers f32 on the CPU of the build
# container): everything but the semantic-similarity test (trained weights) must pass
if [ -d build/synth_model ]; then
  CS_REAL_MODEL_DIR=build/synth_model python -m pytest tests/test_gpu_real_model.py -q -m gpu -k "not semantic" 2>&1 | tail -3 | tee $out/real_model_rehearsal.log
fi
