#!/bin/sh
# see tools/run_rtl_oracle.py
exec python3 "$(dirname "$0")/run_rtl_oracle.py" "$@"
