set -e
for hot in 0 1 2; do
  PROF_KERNEL_REGEX="wave_tile_kernel|permute_x|sum_partials|tile_fixup" bash tools/profile_kernel.sh rmat26_hot$hot --format csr_bal --matrix rmat --n 26 --variants 1 --hot $hot
done
