# Builds the design probes (cross-compiles without a GPU); run the binaries on the GPU box: ./tools/probes/<name>
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing -Wno-unused-value"
/opt/rocm/bin/hipcc $F -o lds_probe lds_probe.hip
/opt/rocm/bin/hipcc $F -o mfma_valu_probe mfma_valu_probe.hip
/opt/rocm/bin/hipcc $F -o mfma_power_probe mfma_power_probe.hip
/opt/rocm/bin/hipcc $F -o mfma_reuse_probe mfma_reuse_probe.hip
/opt/rocm/bin/hipcc $F -o mfma_lds_probe mfma_lds_probe.hip
/opt/rocm/bin/hipcc $F -o mfma_rate_probe mfma_rate_probe.hip
/opt/rocm/bin/hipcc $F -o split_form_probe split_form_probe.hip   # round 6: MFMA + LDS-read ceiling of the 2 x 2 wave split (profiles/r06_split_form_probe.txt)
/opt/rocm/bin/hipcc $F -std=c++17 -ffp-contract=off -w -I ../../alpha_zero_amd/csrc -I ../../include -o spg_tile_probe spg_tile_probe.hip   # round 6: tile shape / ring depth / workgroup-shared B of the wave-per-tile convolution (profiles/r06_spg_tile_probe.txt)
# (the round-2 ablation probes conv_pipe_probe.hip / conv19_probe.hip and their frozen kernel copies az_conv_abl.h / az_conv19_abl.h were
# removed in round 4: their results are profiles/r02_conv_ablation.txt and profiles/r03_conv19_ab.txt; sources in git history, commit c1bb06a)
/opt/rocm/bin/hipcc $F -o lds_dma_coresidency_probe lds_dma_coresidency_probe.hip
/opt/rocm/bin/hipcc $F -o lds_bcast_coresidency_probe lds_bcast_coresidency_probe.hip
# ablation / probe builds of the product kernels (patched copies -> libazsp_abl_*.so, libazsp_head_*.so): python make_sp_abl.py ; python make_head_probe.py
