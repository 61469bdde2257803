#!/bin/bash
# Round 6: paired seeds of the hash family at 2k iterations (batched-gather oracle: 5 GPU-minutes per HIP-vs-oracle seed, 10 per oracle-vs-oracle
# pair).  usage: r6_psnr_hash.sh hip A B   (seeds A..B, batches of 4, one file per batch)  |  r6_psnr_hash.sh floor A B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6psnr; mkdir -p $O
MODE=$1; A=$2; B=$3
P="python3 scripts/psnr_parity.py --family hash"
s=$A
while [ $s -le $B ]; do
  e=$((s + 3)); [ $e -gt $B ] && e=$B
  SEEDS=$(python3 -c "print(','.join(str(i) for i in range($s, $e + 1)))")
  if [ "$MODE" = "hip" ]; then
    timeout 1800 $P --mode hip_vs_oracle --seeds $SEEDS --out $O/psnr_parity_r06_hash_hip_vs_oracle_s${s}_${e}.json > $O/psnr_hip_$s.log 2>&1
    tail -1 $O/psnr_hip_$s.log | cut -c1-400
  else
    timeout 3000 $P --mode oracle_noise_floor --seeds $SEEDS --out $O/psnr_parity_r06_hash_oracle_noise_floor_s${s}_${e}.json > $O/psnr_floor_$s.log 2>&1
    tail -1 $O/psnr_floor_$s.log | cut -c1-400
  fi
  s=$((e + 1))
done
