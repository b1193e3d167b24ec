cd "$GRAFT_REPO_ROOT"
for lib in main gpurun_variants/lib_*.so; do
  if [ "$lib" = main ]; then unset V2V_HIP_LIB; else export V2V_HIP_LIB=$PWD/$lib; fi
  echo "== $lib"; bash tools/profile_cfg5.sh cfg5_fused_convlstm_channels_last 2>/dev/null | grep "convlstm_step_kernel\|conv_halo" | cut -c1-110
done
