export OMP_NUM_THREADS=16 MUMPY_TUNING=1
run() { python tools/train_ddp_bench.py --batch 2 --math fp32 --graph --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
run base
MUMPY_WA_BWD_BLOCKS=512 run wa512
MUMPY_WA_BWD_BLOCKS=768 run wa768
MUMPY_WA_BWD_BLOCKS=1024 run wa1024
