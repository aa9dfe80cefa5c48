for a in "" "--ipc 1" ; do timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline $a 2>&1 | grep -o '"ms_per_step": [0-9.]*' || exit 1; done
DD3D_SINGLE_STREAM=1 timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
