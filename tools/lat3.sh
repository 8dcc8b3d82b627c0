for v in "A=1" "HAK_SPINE_PRIO=0" "HAK_LEVEL_TILE=0" "PINNED=0" "MATCH_CTX=0" "HAK_GRAPH=0 HAK_LEVEL_TILE=0"; do echo "$v: $(env $v python3 tools/single_py.py 2>&1 | tail -1)"; done
