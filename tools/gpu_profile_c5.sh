#!/bin/bash
# rocprofv3 evidence for the codon configuration (MFMA-bound kernel family)
bash tools/gpu_profile.sh c5
