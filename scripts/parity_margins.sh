#!/bin/bash
# the measured parity margins of THIS tree (printed by the full-size tests under -s) -> gpurun_out/<tag>/parity_margins.txt; copy into profiles/
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -s -q -k "atari_literal or dmc_native or autocast or amp_gradients" 2>&1 | grep -v "^$" | grep -iv "warning\|warn(\|autocast(enabled" > gpurun_out/r06/parity_margins.txt
tail -3 gpurun_out/r06/parity_margins.txt
