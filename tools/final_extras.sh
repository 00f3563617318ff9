set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/fin
bash tools/pmc_conv.sh fin/pmc_fwd bf16x3 "up_g4.first T18" > gpurun_out/fin/pmc_fwd.log 2>&1
timeout 100 python tools/conv_bench.py --prec=bf16x3 "up_g4.first T18" > gpurun_out/fin/plain_fwd.log 2>&1
for p in bf16 f32; do
  timeout 250 python bench.py --steps 2 --warmup 1 --cpu-baseline off --precision $p > gpurun_out/fin/bench_$p.json 2> gpurun_out/fin/bench_$p.err
done
timeout 250 python bench.py --steps 2 --warmup 1 --cpu-baseline off --fast > gpurun_out/fin/bench_fast.json 2> gpurun_out/fin/bench_fast.err
for f in gpurun_out/fin/bench_*.json; do python -c "
import json,sys; b=json.load(open('$f')); print('$f', round(b['value'],1), round(b['ms_per_step'],1), b['roofline']['kernel'], round(b['roofline']['achieved'],1))"; done
cat gpurun_out/fin/plain_fwd.log | tail -1
