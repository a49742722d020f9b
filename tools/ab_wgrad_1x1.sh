#!/bin/bash
# the 1x1 weight gradients at the training batch: kernel choice (128 x 128 two-stage / 256 x 256 two-stage / 256 x 256 ring) and split counts
export AB_BATCH=16
for s in res4_2a res4_2c res3_2a res3_2c res5_2a res5_2c res2_2a res2_2c c3; do
  python3 tools/ab_wgrad.py $s RTN_WGRAD_X=0,RTN_WGRAD_DMA=2,RTN_WGRAD_BLOCKS=256,RTN_WGRAD_BLOCKS=1024,RTN_WGRAD_BLOCKS=2048 2>&1 | grep -v amdgpu
done
