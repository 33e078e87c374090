for m in none null kernel_null; do ONE_STREAM=1 POKE_MODE=$m python tools/dbg_graph_poke.py 2>&1 | grep -E "poke mode|after poke|repeat|extra" | cut -c1-120; done
