#!/bin/bash
# Un-profiled per-kernel contributions at the final round-3 build (entry points as the step calls them now):
#   gpurun --timeout 2400 -- 'bash tools/ablate_r03_final.sh'
set -u
O=gpurun_out/r03p
mkdir -p $O
COMMON="nemo_phase_embed_fwd_begin nemo_phase_embed_bwd_colsum nemo_adam_step_dev nemo_kp_bwd_ex nemo_kp_fwd nemo_gmm_fwd_bwd nemo_fk_bwd nemo_fk_fwd nemo_v2v_fused nemo_pose_bwd_fused nemo_kl_fwd_bwd nemo_rot6d_fwd nemo_v2v_prep_fwd_dec nemo_kp_finalize nemo_gemm_grouped_f32"
bash tools/ablate.sh 1 $COMMON nemo_gemm_f32@300x207x20670 nemo_gemm_f32@301x1000x1000 > $O/ablate_v1.txt 2>&1
bash tools/ablate.sh 1 nemo_gemm_f32@301x1000x105 nemo_gemm_f32@301x147x1000 nemo_gemm_f32@301x1000x147 nemo_gemm_f32@301x105x1000 nemo_gemm_f32@300x512x63 nemo_gemm_f32@300x64x512 nemo_gemm_f32@300x512x512 nemo_gemm_f32@300x126x512 nemo_gemm_f32@300x792x207 nemo_gemm_f32@300x207x792 nemo_gemm_f32@300x512x64 nemo_gemm_f32@300x63x512 > $O/ablate_v1b.txt 2>&1
bash tools/ablate.sh 8 $COMMON nemo_gemm_f32@2400x207x20670 nemo_gemm_f32@2401x1000x1000 nemo_gemm_f32@1000x1000x2401 nemo_v2v_combine > $O/ablate_v8.txt 2>&1
bash tools/ablate.sh 2 $COMMON nemo_gemm_f32@600x207x20670 nemo_gemm_f32@601x1000x1000 > $O/ablate_v2.txt 2>&1
bash tools/ablate.sh 4 $COMMON nemo_gemm_f32@1200x207x20670 nemo_gemm_f32@1201x1000x1000 nemo_gemm_f32@1000x1000x1201 > $O/ablate_v4.txt 2>&1
tail -n 3 $O/ablate_v1.txt; tail -n 3 $O/ablate_v8.txt
