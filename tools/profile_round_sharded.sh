# The round's evidence session, second gpurun call:  bash tools/profile_round_sharded.sh
#   what ONE rank of the view-sharded layouts runs per sweep on one GPU (kernel trace + PMC passes of the chain / S-chain / pack
#   kernels): c5 x 8 sliced, c4 x 4 sliced and replicated; then the multi-rank bench rehearsals from the plain command (gloo).
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
bash tools/profile_sliced_rank.sh 8 50000 8000 64 --sliced > $O/rank_c5_sliced.log 2>&1
bash tools/profile_sliced_rank.sh 4 20000 4000 32 --sliced > $O/rank_c4_sliced.log 2>&1
bash tools/profile_sliced_rank.sh 4 20000 4000 32 > $O/rank_c4_replicated.log 2>&1
# (default exchange decision: the library's self-test + the bitwise cross-check -> peer stores here; then the collectives forced)
RESNMTF_BENCH_DEVICE=0 RESNMTF_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_gpus2_auto_p2p_one_gpu.json 2> $O/bench_gpus2_auto_p2p_one_gpu.err
RESNMTF_P2P=0 RESNMTF_BENCH_DEVICE=0 RESNMTF_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_gpus2_gloo_one_gpu.json 2> $O/bench_gpus2_gloo_one_gpu.err
RESNMTF_P2P=0 RESNMTF_BENCH_DEVICE=0 RESNMTF_BENCH_BACKEND=gloo RESNMTF_BENCH_SHAPE=6000,2000,64 RESNMTF_SLICE_CHAINS=1 python3 bench.py --gpus 4 --steps 5 --warmup 2 > $O/bench_gpus4_sliced_gloo_one_gpu.json 2> $O/bench_gpus4_sliced_gloo_one_gpu.err
RESNMTF_P2P=1 RESNMTF_BENCH_DEVICE=0 RESNMTF_BENCH_BACKEND=gloo RESNMTF_BENCH_SHAPE=6000,2000,64 RESNMTF_SLICE_CHAINS=1 python3 bench.py --gpus 4 --steps 5 --warmup 2 > $O/bench_gpus4_sliced_p2p_one_gpu.json 2> $O/bench_gpus4_sliced_p2p_one_gpu.err
python3 tools/time_replica_updates.py 8 10000 2000 16 50 > $O/rank_c2x8_replicated.log 2>&1
echo done
