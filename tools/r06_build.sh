#!/bin/bash
# build libgrx_hip.so and the section-profile variant (run from anywhere)
make 2>&1 | grep -E "error|warning: var" ; cd /root/repo && tools/mkvar.sh prof "-DGRX_PROFILE_SECTIONS" 2>&1 | tail -1; python tools/kernel_resources.py 2>/dev/null | grep -i "tree16<true, false>\|tree<true, false>"
