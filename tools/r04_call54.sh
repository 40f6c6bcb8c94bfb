#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
bash tools/pmc_sq.sh rows80 "k_bar_ohlcv_rows<true>|k_bar_ohlcv_rows<false>" env -C $R python tools/shortbars.py 1e9 4 | tail -40
