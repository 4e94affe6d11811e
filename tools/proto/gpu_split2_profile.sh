cd /root/repo
EXTRA_ARGS="--envs 32768 --kernel 6"
./tools/gpu_pmc_sq.sh r5_p2 upper-body-8192-euler $EXTRA_ARGS || exit 1
./tools/gpu_pmc_sq.sh r5_p2 upper-body-8192-rk4 $EXTRA_ARGS || exit 1
export TMPDIR=/tmp; cd /tmp
OUT=/root/repo/gpurun_out/r5_p2
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_upper-body-32768-euler-split2_$C -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload upper-body-8192-euler --envs 32768 --kernel 6 --steps 40 --warmup 8 --repeats 1 --no-graph > /dev/null 2> $OUT/pmc_split2_$C.err; echo "pmc split2 $C rc=$?"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload upper-body-8192-euler --envs 32768 --kernel 6 --steps 100 --warmup 10 --repeats 3 --no-graph > /dev/null 2> $OUT/prof_stats.err; echo "stats rc=$?"
