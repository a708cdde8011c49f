"""The algebra the f16f6 march kernel rests on (CPU, float64): feature_fc, latent_fc and view_fc have no activation
between them (lib/networks/latent_xyzc.py:105-119), so they are ONE linear layer of fc_2's output and the encodings,
    view_w[:, :256] . latent_w[:, :256] . feature_w                      (128 x 256, over net)
    view_w[:, 256:]                                                      (128 x 90, over [viewdir PE | xyz PE])
    view_b + view_w[:, :256] . (latent_w[:, :256] . feature_b + latent_w[:, 256:] . latent_code + latent_b)
which is what nb_pack_f16_kernel (phase 3) and nb_latent_bias_kernel (second block) build."""
import numpy as np
import torch

from neuralbody_amd import synthetic as syn
from oracle import neuralbody_oracle as orc


def test_folded_colour_head_equals_the_three_layers():
    sd = orc.tensor_state_dict(syn.make_weights(3, num_train_frame=7))
    w = {k: (v[..., 0] if v.dim() == 3 else v).double() for k, v in sd.items()}
    rs = np.random.RandomState(0)
    net = torch.from_numpy(np.maximum(rs.randn(256, 50), 0.0))          # relu(fc_2(...)) of 50 samples
    pe = torch.from_numpy(rs.uniform(-1, 1, (90, 50)))                    # [view PE 27 | xyz PE 63]
    code = w["latent.weight"][4]
    # the reference's sequence (latent_xyzc.py:105-119)
    features = w["feature_fc.weight"] @ net + w["feature_fc.bias"][:, None]
    features = torch.cat([features, code[:, None].expand(128, 50)], 0)
    features = w["latent_fc.weight"] @ features + w["latent_fc.bias"][:, None]
    ref = w["view_fc.weight"] @ torch.cat([features, pe], 0) + w["view_fc.bias"][:, None]
    # the folded layer
    Vg, Vpe = w["view_fc.weight"][:, :256], w["view_fc.weight"][:, 256:]
    Lf, Ll = w["latent_fc.weight"][:, :256], w["latent_fc.weight"][:, 256:]
    W3 = Vg @ Lf @ w["feature_fc.weight"]
    b3 = w["view_fc.bias"] + Vg @ (Lf @ w["feature_fc.bias"] + Ll @ code + w["latent_fc.bias"])
    got = W3 @ net + Vpe @ pe + b3[:, None]
    assert W3.shape == (128, 256) and Vpe.shape == (128, 90)
    assert float((got - ref).abs().max()) <= 1e-12 * float(ref.abs().max())
