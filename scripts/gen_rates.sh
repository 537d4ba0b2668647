set -e
for sh in "1 4" "2 4" "2 5" "1 10" "2 10" "1 20" "2 20"; do
  for algo in lane lane_fma; do python scripts/general_rate.py $sh $algo; done
done
for n in 8192 16384 24576 32768 49152; do for sh in "2 10" "2 20"; do for algo in wave lane_fma; do python scripts/general_rate.py $sh $algo $n; done; done; done
