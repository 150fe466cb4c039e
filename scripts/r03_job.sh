#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 300 python scripts/_dbg_combine.py 2>&1 | grep -v amdgpu.ids
