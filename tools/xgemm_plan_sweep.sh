export OMP_NUM_THREADS=16 MUMPY_TUNING=1
run() { python tools/train_ddp_bench.py --batch 2 --math fp32 --graph --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
run base
MUMPY_XG_TARGET32=512 run t32=512
MUMPY_XG_TARGET32=1024 run t32=1024
MUMPY_XG_TARGET32=1536 run t32=1536
MUMPY_XG_MINCHUNKS=2 run minc=2
MUMPY_XG_MINCHUNKS=8 run minc=8
MUMPY_XG_WIDE_AT=48 run wide=48
MUMPY_XG_WIDE_AT=192 run wide=192
MUMPY_XG_TARGET64=768 run t64=768
MUMPY_XG_TARGET64=256 run t64=256
