#!/bin/bash
# dynamic instruction counts of tokens_kernel on bench.py's two tokenizer launches (cfg4's 65 536 rows, the 524 160-row warehouse launch)
cd "$(dirname "$0")/.."
bash tools/pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" tokinsts python $PWD/tools/tok_cfg4_time.py | grep -E "^#|^kernel|tokens_kernel"
bash tools/pmc.sh "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" tokinsts2 python $PWD/tools/tok_cfg4_time.py | grep -E "^kernel|tokens_kernel"
