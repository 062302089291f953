#!/bin/bash
# usage: ab_test.sh lib.so : runs the mode test against an alternative library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp raytracer_amd/librt_amd.so /tmp/librt_keep.so
cp "$1" raytracer_amd/librt_amd.so
timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py -x -q --timeout 300 --timeout-method thread -m gpu -k "pruning_mode_returns or both_box_modes" 2>&1 | tail -2
cp /tmp/librt_keep.so raytracer_amd/librt_amd.so
