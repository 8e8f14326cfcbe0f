#!/bin/bash
# where the tokens kernel's cycles go: active cycles per instruction class, LDS bank conflicts (GPU box)
cd "$(dirname "$0")/.."
bash tools/pmc.sh "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" tokact python $PWD/tools/tok_cfg4_time.py | grep -E "^kernel|tokens_kernel"
bash tools/pmc.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" toklds python $PWD/tools/tok_cfg4_time.py | grep -E "^kernel|tokens_kernel"
bash tools/pmc.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" tokwave python $PWD/tools/tok_cfg4_time.py | grep -E "^kernel|tokens_kernel"
