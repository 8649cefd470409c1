cd $GRAFT_REPO_ROOT
VTACO_HIP_LIB=/root/repo/variants/lib_hb.so timeout 300 python tools/diag_conv.py 64 32 32 2>&1 | grep -v amdgpu
