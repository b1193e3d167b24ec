#!/bin/bash
# round 2 final profiles: every bench workload under rocprofv3 (kernel stats + PMC traffic + SQ counters), ConvLSTM profile
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/profile_all.sh r02b "cfg2_esim_f32_256x32x256x256_bilinear5 cfg2_noise_free cfg2_dataset_style cfg2_u8 cfg3_v2e_f32_256x32x256x256_bilinear5 cfg3_v2e_u8 cfg4_u8_256x41x256x256_sum5 cfg4_pipeline_720p_to_256_41f_sum5 train_u8_12x201x128x128_sum5"
bash tools/profile_convlstm.sh r02b > gpurun_out/prof_r02b/convlstm.log 2>&1
tail -3 gpurun_out/prof_r02b/convlstm.log
