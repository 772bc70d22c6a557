"""Minimal attribute-dict for options (the reference passes ``addict.Dict`` objects, utils.py:108-208).

Any object with attribute access works for the modules here (an ``addict.Dict`` from the reference's
``load_opts`` included); ``Opts`` exists so this package has no dependency on addict.  ``default_opts()``
restates the subset of ``shared/trainer/defaults.yaml`` that the hot path reads.
"""


class Opts(dict):
    def __init__(self, *a, **kw):
        super().__init__()
        for k, v in dict(*a, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, Opts):
            return Opts(v)
        if isinstance(v, (list, tuple)):
            return type(v)(Opts._wrap(i) for i in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Opts._wrap(v))

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        try:
            return self[k]
        except KeyError:
            # addict auto-vivifies; the reference relies on it in one place (generator.py:144)
            v = Opts()
            super().__setitem__(k, v)
            return v

    def __setattr__(self, k, v):
        self[k] = v


def default_opts() -> Opts:
    """Hot-path subset of shared/trainer/defaults.yaml (line numbers of the reference file in comments)."""
    return Opts({
        "output_path": "output",                                         # :2
        "load_paths": {"p": "none", "m": "none", "pm": "none"},          # :11-14
        "tasks": ["d", "s", "m", "p"],                                   # :19
        "data": {"transforms": [{"name": "resize", "new_size": {"default": 640, "d": 160, "s": 160}}]},   # :61-67
        "gen": {
            "opt": {"optimizer": "ExtraAdam", "beta1": 0.9, "lr": {"default": 0.00005}, "lr_policy": "step",
                    "lr_step_size": 5, "lr_milestones": 15, "lr_gamma": 0.5},   # :73-88
            "encoder": {"architecture": "deeplabv3", "init_type": "xavier", "init_gain": 0.02},   # :92-93,103
            "deeplabv3": {"backbone": "resnet", "output_stride": 8},     # :115-116
            "d": {"architecture": "dada", "upsample_featuremaps": True, "output_dim": 1, "norm": "batch",
                  "init_type": "xavier", "init_gain": 0.02},             # :92-93,122-134
            "s": {"use_advent": True, "use_dada": True, "use_minent": True, "architecture": "deeplabv3", "output_dim": 11,
                  "num_classes": 11, "init_type": "xavier", "init_gain": 0.02},                                    # :135-143
            "m": {"init_type": "xavier", "init_gain": 0.02, "use_advent": True, "use_spade": False, "output_dim": 1, "use_low_level_feats": True,
                  "use_dada": False, "use_pl4m": False, "use_minent": True, "use_minent_var": True, "use_ground_intersection": True, "proj_dim": 64, "n_res": 3, "n_upsample": 3, "norm": "spectral",
                  "activ": "lrelu", "pad_type": "reflect", "use_proj": True,
                  "spade": {"latent_dim": 128, "detach": False, "cond_nc": 15, "spade_use_spectral_norm": True,
                            "spade_param_free_norm": "batch", "num_layers": 3,
                            "activations": {"all_lrelu": True}}},       # :166-190 (+ default-gen :89-99)
            "p": {                                                       # :144-165
                "init_type": "xavier", "init_gain": 0.02, "loss": "gan",
                "latent_dim": 640, "no_z": True, "output_dim": 3, "paste_original_content": True, "pl4m_epoch": 49,
                "spade_kernel_size": 3, "spade_n_up": 7, "spade_param_free_norm": "instance",
                "spade_use_spectral_norm": True, "use_final_shortcut": False,
                "diff_aug": {"use": False, "do_color_jittering": False, "do_cutout": False, "cutout_ratio": 0.5,
                             "do_translation": False, "translation_ratio": 0.125},     # :158-164 (use: true raises)
            },
        },
        "dis": {
            "soft_shift": 0.2, "flip_prob": 0.05,                        # :194-195
            "opt": {"optimizer": "ExtraAdam", "beta1": 0.5, "lr": {"default": 0.00002}, "lr_policy": "step",
                    "lr_step_size": 15, "lr_milestones": 5, "lr_gamma": 0.5},   # :196-211
            "p": {"init_type": "xavier", "init_gain": 0.02, "input_nc": 3, "ndf": 64, "n_layers": 4, "norm": "instance", "use_sigmoid": False, "num_D": 3,
                  "get_intermediate_features": True, "use_local_discriminator": False},   # :213-227
            "m": {"architecture": "base", "gan_type": "WGAN_norm", "init_type": "xavier", "init_gain": 0.02},   # :229-235
            "s": {"gan_type": "WGAN_norm", "init_type": "xavier", "init_gain": 0.02},                              # :236-240
        },
        "events": {"smog": {"airlight": 0.76, "beta": 2, "vr": 1, "yellow_color": [224, 192, 29], "alpha": 20},
                   "fire": {"kernel_size": 281, "kernel_sigma": 140.5, "transparency": 200, "sky_inc_factor": 0.12,
                            "contrast_factor": 1.5, "brightness_factor": 0.95, "crop_bottom_sky_mask": True}},
        # shared/trainer/events.yaml:1-14
        "val": {"val_painter": "none"},   # :323 (a cluster path in the reference: "none" here = no validation painter)
        "train": {"save_n_epochs": 25, "min_save_epoch": 28, "resume": False,   # :313-315
                  "lambdas": {"advent": {"ent_main": 0.5, "ent_aux": 0.0, "ent_var": 0.1, "adv_main": 1.0,
                                         "adv_aux": 0.0, "dis_main": 1.0, "dis_aux": 0.0},      # :303-310
                              "G": {"d": {"main": 1, "gml": 0.5},
                                    "s": {"crossent": 1, "crossent_pseudo": 0.001, "minent": 0.001, "advent": 0.001},
                                    "m": {"bce": 1, "tv": 1, "gi": 0.05, "pl4m": 1},               # :280-292
                                    "p": {"context": 0, "dm": 1, "featmatch": 10, "gan": 1, "reconstruction": 0,
                                          "tv": 0, "vgg": 10}}}},       # :293-300
    })
