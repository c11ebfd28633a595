#!/bin/bash
# round 6, probe 7: the hardware questions of the round-5 review (LDS-DMA rate by issuing waves beside bf16 MFMAs; fp32 MFMA vs
# VALU issue), the short soak tests, the training step's kernel table
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 120 $R/tools/_abl/dma_loaders_mfma > $OUT/r06_ubench_dma_loaders_mfma.txt 2>&1; cat $OUT/r06_ubench_dma_loaders_mfma.txt
timeout 120 $R/tools/_abl/mfma_valu_overlap > $OUT/r06_ubench_mfma_valu_overlap.txt 2>&1; tail -20 $OUT/r06_ubench_mfma_valu_overlap.txt
timeout 120 $R/tools/_abl/dma_rows > $OUT/r06_ubench_dma_rows.txt 2>&1; tail -12 $OUT/r06_ubench_dma_rows.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "soak or rrtmil or bounded" 2>&1 | tail -4 > $OUT/r06_p7_tests.txt; cat $OUT/r06_p7_tests.txt
bash $R/tools/prof_train.sh r06 9000 30
