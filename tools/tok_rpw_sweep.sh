#!/bin/bash
# tokens kernel, rows per wavefront 4 / 8 / 16 on cfg4's own launch and its neighbours (GPU box)
for r in 0 4 8 16; do echo "== MGPT_TOK_RPW=$r"; MGPT_TOK_RPW=$r python tools/tok_cfg4_time.py 2>&1 | grep -v amdgpu.ids; done
