# Builds the design probes (cross-compiles without a GPU); run the binaries on the GPU box: ./tools/probes/<name>
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing -Wno-unused-value"
/opt/rocm/bin/hipcc $F -o lds_probe lds_probe.hip
/opt/rocm/bin/hipcc $F -o mfma_valu_probe mfma_valu_probe.hip
/opt/rocm/bin/hipcc $F -o mfma_power_probe mfma_power_probe.hip
/opt/rocm/bin/hipcc $F -o mfma_reuse_probe mfma_reuse_probe.hip
/opt/rocm/bin/hipcc $F -o mfma_lds_probe mfma_lds_probe.hip
for v in FULL NO_DMA DMA_MASK0 DMA_SAMESRC NO_EPI NO_STORE NO_RESLOAD NO_FRAG HALF_FRAG; do
  D=""; [ $v != FULL ] && D="-DCP_ABL_$v"
  /opt/rocm/bin/hipcc $F $D -o conv_pipe_probe_$v conv_pipe_probe.hip
done
/opt/rocm/bin/hipcc $F -DCP_ABL_NO_DMA -DCP_ABL_NO_EPI -DCP_ABL_NO_RESLOAD -o conv_pipe_probe_MFMA_FRAG conv_pipe_probe.hip
/opt/rocm/bin/hipcc $F -DCP_ABL_NO_DMA -DCP_ABL_NO_EPI -DCP_ABL_NO_RESLOAD -DCP_ABL_NO_FRAG -o conv_pipe_probe_MFMA_ONLY conv_pipe_probe.hip
for v in FULL NO_FRAG NO_DMA; do
  D=""; [ $v != FULL ] && D="-DCP_ABL_$v"
  /opt/rocm/bin/hipcc $F $D -o conv19_probe_$v conv19_probe.hip
done
