"""Free-running Modular streams (tools/synth_free.h) shared by the oracle regression test (CPU) and the HIP parity test (GPU)."""
import synth_lib as S

FREE_CASES = {
    # name: kwargs of synth_lib.encode_modular_free (tools/synth_free.h)
    "random_tree_small": dict(seed=3),
    "random_tree_all_predictors": dict(seed=4, tree_flags=S.TREE_ALL_PREDICTORS | S.TREE_MULTIPLIERS, w=200, h=150),
    "weighted_predictor_custom_header": dict(seed=5, tree_flags=S.TREE_WP | S.TREE_ALL_PREDICTORS | S.TREE_MULTIPLIERS | S.TREE_CUSTOM_WP, w=300, h=280),
    "previous_channel_properties_global": dict(seed=6, tree_flags=S.TREE_PREV_CHANNELS | S.TREE_ALL_PREDICTORS, w=120, h=90, has_alpha=True),
    "previous_channel_properties_groups": dict(seed=7, tree_flags=S.TREE_PREV_CHANNELS | S.TREE_WP, w=520, h=300, has_alpha=True),
    "local_trees_in_sections": dict(seed=8, local_trees=1, w=600, h=300, tree_flags=S.TREE_WP | S.TREE_ALL_PREDICTORS),
    "local_tree_everywhere": dict(seed=9, local_trees=2, w=300, h=520, tree_flags=S.TREE_PREV_CHANNELS),
    "local_tree_single_group": dict(seed=10, local_trees=2, w=100, h=80),
    "lz77_global": dict(seed=11, lz77=True, w=200, h=120, tree_flags=S.TREE_ALL_PREDICTORS),
    "lz77_sections_wp": dict(seed=12, lz77=True, w=530, h=270, tree_flags=S.TREE_WP | S.TREE_MULTIPLIERS),
    "lz77_local_trees": dict(seed=13, lz77=True, local_trees=1, w=300, h=300),
    "palette_plain_global": dict(seed=14, palette=True, nb_colors=20, w=150, h=100),
    "palette_delta_gradient": dict(seed=15, palette=True, nb_colors=24, nb_deltas=5, pal_pred=5, w=150, h=100),
    "palette_delta_wp_sections": dict(seed=16, palette=True, nb_colors=40, nb_deltas=7, pal_pred=6, w=400, h=300, has_alpha=True, tree_flags=S.TREE_WP),
    "palette_delta_every_predictor": dict(seed=17, palette=True, nb_colors=12, nb_deltas=3, pal_pred=13, w=90, h=70),
    "gray_alpha_16bit_everything": dict(seed=18, nchan=1, has_alpha=True, tree_flags=31, lz77=True, local_trees=1, w=300, h=270),
}
