export SA_GUARD=0
SA_KERNEL_DEFINES=-DSA_ABLATE_PROFILE python tools/profile_lv.py 65536 lv 2>&1 | tail -4
SA_KERNEL_DEFINES=-DSA_ABLATE_PROFILE python tools/profile_lv.py 262144 robertson 2>&1 | tail -4
