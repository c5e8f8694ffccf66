#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
tools/ab.sh r04_ab8.txt "KNP_EMI_CHEB=0" "KNP_EMI_CHEB=0 KNP_SETUP_NATIVE_GRAM=0" "KNP_NOP=1" "KNP_SETUP_NATIVE_GRAM=0"
