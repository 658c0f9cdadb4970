mkdir -p gpurun_out/r3e
python tools/conv_mix.py "32,56,2001 64,28,2001 128,14,2020 256,7,0" "32,56,2001 64,28,2012 128,14,2020 256,7,0" "32,56,2012 64,28,2012 128,14,2020 256,7,0" "32,56,2001 64,28,2012 128,14,2020 256,7,2020" "32,56,2001 64,28,2012 128,14,2020" 2>&1 | grep -E "===|conv_micro" | sed 's/\[conv_micro\] //;s/k 3 s 1 //;s/dbg 0: //'
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 > gpurun_out/r3e/bench_$tag.json
  python -c "
import json;d=json.loads(open('gpurun_out/r3e/bench_$tag.json').read());print('$tag:',d['value'],d['ms_per_step'])"; }
run w4s5 GRNET_WINO4S=5
run w4s5_w4r2 GRNET_WINO4S=5 GRNET_WINO4R=2
run w4s7_w4r2 GRNET_WINO4R=2
