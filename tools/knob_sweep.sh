cd "${GRAFT_REPO_ROOT:-.}"
run() { v=$(env "$@" python bench.py --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"); echo "$* -> $v"; }
run X=1
run C3D_WG_BLOCKS=192
run C3D_WG_BLOCKS=384
run C3D_WG_BLOCKS=512
run C3D_PW_FORCE8=2
run C3D_DWBD_TPW=4
run C3D_DWBD_TPW=8
run C3D_DWBD_TPW=16
run C3D_DW_TPW=4
run C3D_DW_TPW=8
run C3D_DW_TPW=16
run C3D_BOB_GRID=256
run C3D_BOB_GRID=512
run C3D_WG_MT=64
run X=2
