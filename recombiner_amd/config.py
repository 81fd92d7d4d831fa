"""Modality presets -- same keys and values as the reference's config.py:28-137 (the call surface
`configs[dataset]` is kept verbatim so existing drivers work unchanged)."""


def _preset(input_dim, output_dim, data_dim, pixel_sizes, patch, hierarchical_patch_nums, patch_nums,
            fourier_dim, layerwise_scale_factors, upsample_factors, bitrate_range, lowest_bitrate):
    return {
        "input_dim": input_dim, "output_dim": output_dim, "hidden_dims": [32] * 3,
        "data_dim": data_dim, "pixel_sizes": pixel_sizes, "patch": patch,
        "hierarchical_patch_nums": hierarchical_patch_nums, "patch_nums": patch_nums,
        "latent_dim": 128, "fourier_dim": fourier_dim, "paddings": [2, 1, 1],
        "layerwise_scale_factors": layerwise_scale_factors, "upsample_factors": upsample_factors,
        "bitrate_range": bitrate_range, "lowest_bitrate": lowest_bitrate,
    }


configs = {
    "cifar": _preset(32, 3, 2, [32, 32], False, None, None, 16, [4, 2, 2], [16, 16], 0.3, 0.1),
    "kodak": _preset(32, 3, 2, [64, 64], True, {"level2": [4, 4], "level3": [8, 12]}, [512 // 64, 768 // 64],
                     16, [4, 2, 2], [16, 16], 0.1, 0.05),
    "audio": _preset(32, 1, 1, [800], True, {"level2": [4], "level3": [60]}, [48000 // 800],
                     16, [4, 2, 2], [16], 0.3, 0.1),
    "video": _preset(34, 3, 3, [24, 16, 16], True, {"level2": [1, 4, 4], "level3": [1, 8, 8]},
                     [24 // 24, 128 // 16, 128 // 16], 18, [(6, 4, 4), 2, 2], [24, 16, 16], 0.3, 0.1),
    "protein": _preset(32, 3, 1, [96], False, None, None, 16, [4, 2, 2], [16], 0.3, 0.1),
}
