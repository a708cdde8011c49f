"""Plugin files for the reference's `imp.load_source(cfg.X_module, cfg.X_path)` factories
(lib/networks/make_network.py:5-9, lib/networks/renderer/make_renderer.py:5-9,
lib/train/trainers/make_trainer.py:5-14).  Select them from the command line, e.g.

    python run.py --type visualize --cfg_file configs/zju_mocap_exp/latent_xyzc_313.yaml \
        network_path /path/to/neuralbody_amd/plugins/latent_xyzc.py \
        renderer_path /path/to/neuralbody_amd/plugins/if_clight_renderer.py \
        trainer_path /path/to/neuralbody_amd/plugins/if_nerf_clight.py
"""
