#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/t2.txt
