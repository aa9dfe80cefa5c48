#!/bin/bash
# A/B of launch knobs under the final schedule (kernels unchanged; every knob is read once per process): ms per step, 300 steps each.
run() { local a="$1"; shift; echo -n "$* [$a] : "; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4))" || exit 1; }
for w in 0 32 48 64 80 96 128; do run "--steps 300" DD3D_ARP_WAVE_ROWS=$w; done
for w in 64 128; do run "--steps 200 --config cfg3" DD3D_ARP_WAVE_ROWS=$w; run "--steps 300 --emulate-shard 0/8" DD3D_ARP_WAVE_ROWS=$w; run "--steps 300 --emulate-shard 2/8" DD3D_ARP_WAVE_ROWS=$w; run "--steps 5 --warmup 2 --config cfg4" DD3D_ARP_WAVE_ROWS=$w; done
