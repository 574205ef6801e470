#!/bin/bash
# The timing-only and A/B variants of kernels.hip that scripts/collect_profiles.sh's probe targets load (scripts/probe_build/k1v_<name>.so,
# git-ignored, travel with the gpurun snapshot).  hipcc cross-compiles: run this HERE before the gpurun call.
set -e
cd "$(dirname "$0")/.."
python scripts/k1_variants.py --build base= \
  q1=-DHYDK_CHAIN_PROBE=1 q4=-DHYDK_CHAIN_PROBE=4 q8=-DHYDK_CHAIN_PROBE=8 q16=-DHYDK_CHAIN_PROBE=16 \
  k1c0=-DHYDK_K1_PRIO=1,-DHYDK_CHAIN_PRIO=0 k2c0=-DHYDK_K1_PRIO=2,-DHYDK_CHAIN_PRIO=0 k3c0=-DHYDK_K1_PRIO=3,-DHYDK_CHAIN_PRIO=0 k3c2=-DHYDK_K1_PRIO=3,-DHYDK_CHAIN_PRIO=2 \
  ls1=-DHYDK_LANE_STEP=1 ls2=-DHYDK_LANE_STEP=2 \
  q32=-DHYDK_CHAIN_PROBE=32 hog=-DHYDK_CHAIN_HOG=1 \
  r5=-DHYDK_LANE_PIPE=0,-DHYDK_LANE_STEP=1 pp1=-DHYDK_LANE_PIPE=1 pp2=-DHYDK_LANE_PIPE=2 pp2s2=-DHYDK_LANE_PIPE=2,-DHYDK_LANE_STEP=2 \
  gt=-DHYDK_LANE_TAB_GLOBAL=1 gt1=-DHYDK_LANE_TAB_GLOBAL=1,-DHYDK_LANE_PIPE=1 \
  pp1n7=-DHYDK_LANE_PIPE=1,-DHYDK_LANE_NC9_PROBE=7 pp1n4=-DHYDK_LANE_PIPE=1,-DHYDK_LANE_NC9_PROBE=4 pp2n7=-DHYDK_LANE_NC9_PROBE=7 pp2n4=-DHYDK_LANE_NC9_PROBE=4 \
  w5=-DHYDK_K1_WAVES=5 w5i1=-DHYDK_K1_WAVES=5,-DHYDK_K1_ILP=1 i1=-DHYDK_K1_ILP=1 \
  cs=-DHYDK_K1_CHANSEQ=1 cs1=-DHYDK_K1_CHANSEQ=1,-DHYDK_LANE_PIPE=1
