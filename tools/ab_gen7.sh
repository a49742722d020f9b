#!/bin/bash
# generation 7 (filters in registers) against generation 4 on the layers both take: same process, interleaved (tools/ab_conv.py)
RTN_CONV_H8R=1 python3 tools/ab_conv.py pyramid_regression_1,pyramid_classification_0,P3,P4,res4b_branch2b RTN_CONV_H8R=1,RTN_CONV_H8R=0,RTN_CONV_H8R=1+RTN_CONV_H8_MI=3,RTN_CONV_H8R=0+RTN_CONV_H8_MI=3
