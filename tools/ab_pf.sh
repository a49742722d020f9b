#!/bin/bash
# the prefetching step of generation 4 (RTN_CONV_H8_PF) against the plain step, same process, interleaved (tools/ab_conv.py)
python3 tools/ab_conv.py pyramid_regression_1,pyramid_classification_0,P3,P4,res4b_branch2b RTN_CONV_H8_PF=0,RTN_CONV_H8_PF=2,RTN_CONV_H8_PF=0+RTN_CONV_H8_MI=3,RTN_CONV_H8_PF=2+RTN_CONV_H8_MI=3
