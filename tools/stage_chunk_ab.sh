#!/bin/bash
# the drop-in path of the unmodified reference prover (oracle/_ref/plonk_gpu, 2^16 gates, third proof of the process) for different sizes of the
# staging chunks the callers' buffers cross the link in (BBGPU_STAGE_CHUNK_BYTES; four pinned buffers of 4 MiB form the ring), and the
# host-buffer transform at 2^18 / 2^20 beside it
export OMP_NUM_THREADS=16 BBGPU_SHIM_STRICT=1 BB_WARM_PROOFS=2
for ch in 4194304 2097152 1048576 524288 262144; do
  for rep in 1 2; do
    echo -n "chunk $ch: "; BBGPU_STAGE_CHUNK_BYTES=$ch oracle/_ref/plonk_gpu prove 65536 2>&1 >/dev/null | grep construct_proof
  done
done
for th in 0 1 3 7; do echo -n "chunk 1 MiB, copy helpers $th: "; BBGPU_STAGE_THREADS=$th BBGPU_STAGE_CHUNK_BYTES=1048576 oracle/_ref/plonk_gpu prove 65536 2>&1 >/dev/null | grep construct_proof; done
