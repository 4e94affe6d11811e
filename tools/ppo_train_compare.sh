#!/bin/bash
# PPO training run of 40 M timesteps at 4 096 envs on one MI355X: fused policy kernels (default) against the torch path
cd /root/repo
for mode in fused torch; do
  rm -rf /tmp/res_$mode
  flag=""; [ $mode = torch ] && flag="--torch-policy"
  t0=$(date +%s.%N)
  timeout -k 10 400 python -m gym_roboy_amd.train_parallel 4096 /tmp/res_$mode --rounds 1 --steps-per-round 40000000 $flag > gpurun_out/ppo_train_$mode.log 2>&1
  rc=$?
  t1=$(date +%s.%N)
  echo "mode=$mode rc=$rc wall $(python -c "print(round($t1 - $t0, 1))") s (process start included)"
  grep mean_reward gpurun_out/ppo_train_$mode.log | sed -n '1p;$p' | cut -c1-200
done
