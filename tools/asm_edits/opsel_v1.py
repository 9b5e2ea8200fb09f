# variants of the failing packed add (same values): 1 sources swapped (op_sel on src0), 2 operand swizzled by two v_mov in front (no op_sel),
# 3 as a packed fma x * 1.0 + y with op_sel on src2, 4 the original instruction issued TWICE into different destinations, first result kept
V = 1
for k, l in enumerate(K):
    if "v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]" in l:
        if V == 1: K[k] = "\tv_pk_add_f32 v[36:37], v[4:5], v[36:37] op_sel:[1,0]"
        if V == 2: K[k] = "\tv_mov_b32_e32 v70, v5\n\tv_mov_b32_e32 v71, v5\n\tv_pk_add_f32 v[36:37], v[36:37], v[70:71]"
        if V == 3: K[k] = "\tv_pk_fma_f32 v[36:37], v[36:37], 1.0, v[4:5] op_sel:[0,0,1] op_sel_hi:[1,0,1]"
        if V == 4: K[k] = "\tv_pk_add_f32 v[70:71], v[36:37], v[4:5] op_sel:[0,1]\n\tv_pk_add_f32 v[72:73], v[36:37], v[4:5] op_sel:[0,1]\n\tv_mov_b32_e32 v36, v70\n\tv_mov_b32_e32 v37, v71"
