import synth_lib as S


def enc(seed, w, h, shape=1, mix=1, epf=1):
    S.set_lf_tree_shape(shape)
    try:
        return S.encode_vardct(S.synthetic_image(seed, w, h), seed=seed, strategy_mix=mix, epf_iters=epf)
    finally:
        S.set_lf_tree_shape(0)
