cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 2 3; do
  L=exp/lib_abl$v.so; [ $v = 0 ] && L=pilco_amd/libpilco_hip.so
  rm -rf gpurun_out/abl$v; mkdir -p gpurun_out/abl$v
  PILCO_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl$v -o r -- python tools/grad_ab.py save /tmp/x.npz > gpurun_out/abl$v/log 2>&1 </dev/null
  echo "REV_ABL=$v: $(grep k_rev_step gpurun_out/abl$v/r_kernel_stats.csv | cut -d, -f1-4)"
done
