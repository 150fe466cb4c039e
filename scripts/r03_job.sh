#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 300 python scripts/_dbg_hiccup.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
