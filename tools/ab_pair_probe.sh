#!/bin/bash
# On the GPU box: the channel-split wave-pair probe (timing only, results wrong) against the shipped H = 32 backward.
mkdir -p gpurun_out
AB_ROUNDS=3 timeout 900 python3 tools/ab.py base=pointvs_amd/libpvs_egnn.so pair3=pointvs_amd/libpvs_egnn_pairprobe.so pair2=pointvs_amd/libpvs_egnn_pairprobe2.so > gpurun_out/ab_pair_probe.txt 2>&1
cat gpurun_out/ab_pair_probe.txt
