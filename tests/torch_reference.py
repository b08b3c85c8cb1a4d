"""TEST INFRASTRUCTURE -- plain-PyTorch restatement of the reference encoder forward
(gennbv/network/hybrid_encoder.py:76-98), generalised to a grid edge G, as the fp32 / fp64
reference of the floating-point kernels.  Proven equal to the reference's own modules on fixtures
F7 / F9 (tests/test_policy_ppo_cpu.py), which makes it a legitimate oracle at the grid sizes the
reference hard-codes away (G != 20).  Never imported by the product package."""
import torch

from gennbv_amd.network.hybrid_encoder import Hybrid_Encoder


class TorchHybridEncoder(Hybrid_Encoder):
    """Same parameters / state_dict keys as the product class; forward through torch's library kernels."""
    backend = "torch"

    def forward(self, observations) -> torch.Tensor:
        if not isinstance(observations, torch.Tensor):  # ops.encoder_ops.RowGather / DenseObs
            observations = observations.materialize()
        num_env = observations.shape[0]
        s = self.state_input_shape[0]
        g = self.grid_size
        action_input = observations[:, :s].view(num_env, -1, 6)
        action_input = self.positional_encoding(action_input).view(num_env, -1)
        grid_input = observations[:, s:s + g ** 3].reshape(num_env, 1, g, g, g)
        feature_action = self.naive_encoder_action(action_input)
        feature_grid = self.naive_encoder_grid(grid_input).reshape(num_env, -1)
        feature_grid = self.output_layer_grid(feature_grid)
        if self.semantic_branch:  # (opt-in, build-defined: network/hybrid_encoder.py)
            rgb = observations[:, s + g ** 3:s + g ** 3 + 8192]
            emb = self.naive_encoder_rgb(self.rgb_patches(rgb)).reshape(num_env, -1)
            return self.output_layer(torch.cat((feature_action, feature_grid, self.output_layer_rgb(emb)), dim=-1))
        return self.output_layer(torch.cat((feature_action, feature_grid), dim=-1))


def encoder_class(backend: str):
    return {"torch": TorchHybridEncoder, "hip": Hybrid_Encoder}[backend]
