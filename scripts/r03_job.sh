#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 400 python scripts/_dbg_leak.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
