"""ACT (action-chunking transformer, CVAE) with the point-cloud tokenizer -- ``ACTPCD``.

Behavioural counterpart of /root/reference/src/models/components/act/act.py:40-309 (ACT) and
:312-598 (ACTPCD): same constructor arguments, same ``forward(data_dict) -> data_dict`` contract
(keys ``a_hat``, ``is_pad_hat``, ``mu``, ``logvar``, ``action_loss``, ``kl_loss``, ``loss``) and the
same parameter names, so Hydra ``_target_`` configs and reference checkpoints keep working.

What is different is the execution plan, not the maths:
* activations are batch-first end to end and attention is fused (policy/transformer.py);
* farthest point sampling and the kNN query depend only on coordinates (act.py:395,447 use ``p``
  alone for the indices), so they are issued on a side HIP stream and overlap the PointNet MLP;
* the set-abstraction layer (act.py:384-465: FPS -> kNN -> group -> Linear -> BN -> ReLU -> max)
  can run either in the reference's op order (``sa_impl="reference"``) or through the fused HIP
  kernels of policy/sa_fused.py (``sa_impl="fused"``);
* no per-cloud host synchronisation: offsets carry a host copy (pointops._common.host_offsets).

``pointops`` is injected (default: the HIP package).  Tests and the CPU baseline inject the CPU
oracle's module; the product never imports it.
"""
import numpy as np
import torch
import torch.nn as nn

from .sa_layer import coord_embedding_sine, set_abstraction


def get_sinusoid_encoding_table(n_position, d_hid):
    """act/utils.py:42-55: (1, n_position, d_hid) float32, float64 angle arithmetic."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    table = pos / np.power(10000, 2 * (j // 2) / d_hid)
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.FloatTensor(table).unsqueeze(0)


def reparametrize(mu, logvar, eps=None):
    """act/utils.py:36-39.  ``eps`` can be injected for reproducible parity tests."""
    std = logvar.div(2).exp()
    if eps is None:
        eps = torch.randn_like(std)
    return mu + std * eps



def _lin(module, x):
    """module(x) for the policy's small nn.Linear layers through rows_linear.linear_rows: same forward; in training on the GPU its
    backward forms the bias gradient with the library's column sums instead of the framework's bf16 reduction (rows_linear.bias_grad)."""
    from .rows_linear import linear_rows

    return linear_rows(x, module.weight, module.bias)


class ACTPCD(nn.Module):
    def __init__(
        self,
        backbone,
        transformer,
        encoder,
        hidden_dim,
        num_queries,
        num_cameras=0,
        action_dim=8,
        qpos_dim=9,
        env_state_dim=0,
        latent_dim=32,
        action_loss=None,
        klloss=None,
        kl_weight=20.0,
        goal_cond_dim=0,
        obs_feature_pos_embedding=None,
        freeze_backbone=False,
        pcd_nsample=16,
        pcd_npoints=1024,
        sampling="fps",
        heatmap_th=0.1,
        ignore_vae=False,
        use_mask=False,
        bg_ratio=0.0,
        pre_sample=False,
        in_channels=6,
        pointops=None,
        sa_impl="reference",
        overlap_sampling=True,
        dead_decoder_layers="keep",
    ):
        super().__init__()
        if backbone is None:
            raise ValueError("ACTPCD needs a point-cloud backbone")
        if not 0.0 <= bg_ratio < 1.0:
            raise ValueError("bg_ratio must be in [0, 1)")
        if "fps" not in sampling:
            raise NotImplementedError(sampling)
        if pointops is None:
            from .. import pointops as _hip_pointops

            pointops = _hip_pointops
        self._pointops = [pointops]  # in a list: not a submodule / not in the state dict
        self.sa_impl = sa_impl
        # act.py:270 reads only the first decoder layer's output; see TransformerDecoder.first_only for the three ways
        # to treat the other six ("keep" = the reference's autograd graph, the default)
        if dead_decoder_layers not in ("keep", "prune_backward", "skip"):
            raise ValueError(dead_decoder_layers)
        self.dead_decoder_layers = dead_decoder_layers
        transformer.decoder.first_only = dead_decoder_layers
        self.overlap_sampling = overlap_sampling

        self.backbone = backbone
        self.transformer = transformer
        self.encoder = encoder
        self.num_queries = num_queries
        self.num_cameras = 0
        self.action_dim = action_dim
        self.qpos_dim = qpos_dim
        self.env_state_dim = env_state_dim
        self.hidden_dim = hidden_dim
        self.kl_weight = kl_weight
        self.latent_dim = latent_dim
        self.goal_cond_dim = goal_cond_dim
        self.freeze_backbone = freeze_backbone
        self.ignore_vae = ignore_vae
        if freeze_backbone:
            for p in self.backbone.parameters():
                p.requires_grad = False
        self.action_loss = action_loss if action_loss is not None else nn.MSELoss(reduction="none")
        self.klloss = klloss

        # ---- ACT.build_encoder (act.py:93-122); input_proj is set to None by ACTPCD (:361)
        self.input_proj = None
        self.input_proj_robot_state = nn.Linear(qpos_dim, hidden_dim)
        self.cls_embed = nn.Embedding(1, hidden_dim)
        self.encoder_action_proj = nn.Linear(action_dim, hidden_dim)
        self.encoder_joint_proj = nn.Linear(qpos_dim, hidden_dim)
        self.latent_proj = nn.Linear(hidden_dim, latent_dim * 2)
        self.register_buffer("pos_table", get_sinusoid_encoding_table(1 + 1 + num_queries, hidden_dim))
        if goal_cond_dim > 0:
            self.proj_goal_cond_emb = nn.Linear(goal_cond_dim, hidden_dim)
        # ---- ACT.build_decoder (act.py:124-135)
        self.action_head = nn.Linear(hidden_dim, action_dim)
        self.is_pad_head = nn.Linear(hidden_dim, 1)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.latent_out_proj = nn.Linear(latent_dim, hidden_dim)
        self.additional_pos_embed = nn.Embedding(2 + int(goal_cond_dim > 0), hidden_dim)
        # ---- ACTPCD tokenizer (act.py:363-382)
        self.pcd_nsample = pcd_nsample
        self.pcd_npoints = pcd_npoints
        self.pre_sample = pre_sample
        if not pre_sample:
            self.linear = nn.Linear(3 + backbone.num_channels, hidden_dim, bias=False)
            self.bn = nn.BatchNorm1d(hidden_dim)
        else:
            # act.py:369-376: the set-abstraction layer sits IN FRONT of the backbone and keeps the raw feature width
            # (`backbone.in_channels`; the constructor's own `in_channels` argument is never read by the reference either),
            # the backbone then runs on the m sampled points and its output IS the token matrix.  Selected by
            # configs/exp_maniskill2_act_policy/maniskill2_model/scratch_pointnet_pcd_presample{,_wo_rgb,_wo_xyz}.yaml.
            cin = backbone.in_channels
            self.linear = nn.Linear(3 + cin, cin, bias=False)
            self.bn = nn.BatchNorm1d(cin)
        self.pool = nn.MaxPool1d(pcd_nsample)
        self.relu = nn.ReLU(inplace=True)
        self.sampling = sampling
        self.use_mask = use_mask
        self.bg_ratio = bg_ratio
        self._side_stream = None

    @property
    def pointops(self):
        return self._pointops[0]

    # ------------------------------------------------------------------ CVAE encoder (act.py:137-188)
    def forward_encoder(self, data_dict):
        qpos = data_dict["qpos"]
        actions = data_dict.get("actions", None)
        is_pad = data_dict.get("is_pad", None)
        is_training = actions is not None
        bs = qpos.shape[0]
        data_dict["is_training"] = is_training
        if is_training and not self.ignore_vae:
            action_embed = _lin(self.encoder_action_proj, actions)  # (B, T, C)
            qpos_embed = _lin(self.encoder_joint_proj, qpos).unsqueeze(1)  # (B, 1, C)
            cls_embed = self.cls_embed.weight.unsqueeze(0).expand(bs, -1, -1)  # (B, 1, C)
            enc_in = torch.cat([cls_embed, qpos_embed, action_embed], dim=1)  # (B, T+2, C)
            pad = torch.cat([is_pad.new_zeros(bs, 2), is_pad], dim=1)  # CLS / qpos are never padding
            enc_out = self.encoder(enc_in, pos=self.pos_table, src_key_padding_mask=pad)
            latent_info = _lin(self.latent_proj, enc_out[:, 0])  # CLS token
            from . import fused_ops

            eps = data_dict.get("vae_eps", None)
            if fused_ops.cvae_latent_supported(latent_info, self.latent_dim, eps):
                # split + reparametrisation (+ contiguous mu / logvar for the KL term) in one launch each way
                latent_sample, mu, logvar = fused_ops.cvae_latent(latent_info, eps)
            else:
                mu = latent_info[:, : self.latent_dim]
                logvar = latent_info[:, self.latent_dim :]
                latent_sample = reparametrize(mu, logvar, eps)
        else:
            mu = logvar = None
            latent_sample = torch.zeros([bs, self.latent_dim], dtype=torch.float32, device=qpos.device)
        data_dict["mu"] = mu
        data_dict["logvar"] = logvar
        data_dict["latent_input"] = _lin(self.latent_out_proj, latent_sample)
        return data_dict

    # ------------------------------------------------------------------ tokenizer (act.py:384-551)
    def _new_offsets(self, o):
        """n_o = [M, 2M, ...] (act.py:387-391), cached per (b, device) with its host copy attached."""
        b = int(o.shape[0])
        key = (b, o.device)
        cache = self.__dict__.setdefault("_n_o_cache", {})
        if key not in cache:
            host = [self.pcd_npoints * (i + 1) for i in range(b)]
            t = torch.tensor(host, dtype=torch.int32, device=o.device)
            t._pcm_host = host
            cache[key] = t
        return cache[key]

    def pcd_sampling(self, pxo, mask=None, return_index=False):
        """(p (n,3), x (n,c), o (b)) -> [n_p (m,3), x (m,H), n_o (b)] -- the set-abstraction layer."""
        p, x, o = pxo
        n_o = self._new_offsets(o)
        pre = set_abstraction.sample_and_query(self, self.pointops, p, o, n_o, mask=mask)
        out = set_abstraction(self, self.pointops, p, x, o, n_o, impl=self.sa_impl, pre=pre)
        n_p, feat, idx = out
        if return_index:
            return [n_p, feat, n_o, idx]
        return [n_p, feat, n_o]

    def coord_embedding_sine(self, coord, temperature=10000, normalize=False, scale=None):
        return coord_embedding_sine(coord, self.hidden_dim, temperature, normalize, scale)

    def prefetch_sampling(self, pcd_dict):
        """Start FPS + kNN for a future batch's clouds on the side stream (see sa_layer.prefetch_sampling)."""
        coord, offset = pcd_dict["coord"], pcd_dict["offset"]
        set_abstraction.prefetch_sampling(self, self.pointops, coord, offset, self._new_offsets(offset),
                                          mask=self._mask_of(pcd_dict))

    def sampling_for(self, pcd_dict, overlap=True):
        """FPS / kNN / index statistics of these clouds: the prefetched result if `prefetch_sampling` saw them, else
        computed now (on the side stream with `overlap`)."""
        coord, offset = pcd_dict["coord"], pcd_dict["offset"]
        return set_abstraction.sample_and_query(self, self.pointops, coord, offset, self._new_offsets(offset), overlap=overlap,
                                                mask=self._mask_of(pcd_dict))

    def _mask_of(self, pcd_dict):
        """act.py:511-515: with ``use_mask`` the batch must carry the per-point foreground mask (KeyError otherwise)."""
        return pcd_dict["mask"] if self.use_mask else None

    def install_static_sampling(self, pcd_dict, pre):
        set_abstraction.install_static(self, pcd_dict["coord"], pcd_dict["offset"], pre)

    def load_static_sampling(self, pre):
        set_abstraction.load_static(self, pre)

    tokenizer_fp32 = True  # policy/precision.py: the tokenizer stays in fp32 under bf16 autocast

    def tokenizer_modules(self):
        """Modules whose parameters the tokenizer consumes (no bf16 mirror for them while `tokenizer_fp32`)."""
        return [self.backbone, self.linear, self.bn]

    def forward_pcd_embed(self, pcd_dict):
        from .precision import tokenizer_autocast

        with tokenizer_autocast(self, pcd_dict["coord"]):
            return self._forward_pcd_embed(pcd_dict)

    def _forward_pcd_embed(self, pcd_dict):
        coord, offset = pcd_dict["coord"], pcd_dict["offset"]
        n_o = self._new_offsets(offset)
        # indices first (coordinates only), overlapped with the backbone when on the GPU
        pre = set_abstraction.sample_and_query(self, self.pointops, coord, offset, n_o,
                                               overlap=self.overlap_sampling and coord.is_cuda, mask=self._mask_of(pcd_dict))
        if self.pre_sample:
            # act.py:509-530: sample first (on the raw features), then the backbone on the sampled cloud.  Like the
            # reference, the cloud dict handed in is rewritten to describe the sampled cloud.
            n_p, feat, fps_idx = set_abstraction(self, self.pointops, coord, pcd_dict["feat"], offset, n_o, impl=self.sa_impl, pre=pre)
            pcd_dict["coord"], pcd_dict["feat"], pcd_dict["offset"] = n_p, feat, n_o
            pcd_dict["grid_coord"] = pcd_dict["grid_coord"][fps_idx.long()]
            tokens = self.backbone(pcd_dict)
        else:
            features = self.backbone(pcd_dict)
            n_p, tokens, _ = set_abstraction(self, self.pointops, coord, features, offset, n_o, impl=self.sa_impl, pre=pre)
        b = offset.shape[0]
        pcd_pos = coord_embedding_sine(n_p, self.hidden_dim)
        # "(b n) c -> b c 1 n"
        tokens = tokens.view(b, self.pcd_npoints, -1).permute(0, 2, 1).unsqueeze(2)
        pcd_pos = pcd_pos.view(b, self.pcd_npoints, -1).permute(0, 2, 1).unsqueeze(2)
        return tokens, pcd_pos

    # ------------------------------------------------------------------ act.py:553-598
    def forward_obs_embed(self, data_dict):
        qpos = data_dict["qpos"]
        latent_input = data_dict["latent_input"]
        goal_cond = None
        if self.goal_cond_dim > 0:
            if data_dict["goal_cond"].dim() > 2:
                data_dict["goal_cond"] = data_dict["goal_cond"].reshape(data_dict["goal_cond"].shape[0], -1)
            goal_cond = _lin(self.proj_goal_cond_emb, data_dict["goal_cond"])
        if "pcd_embed" in data_dict:  # tokens computed by an earlier stage (BCTrainer mode="hybrid")
            pcd_tokens, pcd_pos = data_dict["pcd_embed"]
        else:
            pcd_tokens, pcd_pos = self.forward_pcd_embed(data_dict["pcds"])
        from . import staging

        pcd_tokens = staging.cut("tokens", pcd_tokens)  # the tokenizer's backward is the last stage
        proprio_input = _lin(self.input_proj_robot_state, qpos).unsqueeze(0)
        if goal_cond is not None:
            proprio_input = torch.cat([proprio_input, goal_cond.unsqueeze(0)], dim=0)
        data_dict["src"] = pcd_tokens
        data_dict["pos"] = pcd_pos
        data_dict["latent_input"] = latent_input.unsqueeze(0)
        data_dict["proprio_input"] = proprio_input
        return data_dict

    # ------------------------------------------------------------------ act.py:255-291
    def forward_decoder(self, data_dict):
        hs = self.transformer(
            data_dict["src"], None, self.query_embed.weight, data_dict["pos"], data_dict["latent_input"],
            data_dict["proprio_input"], self.additional_pos_embed.weight,
        )[0]  # only the FIRST decoder layer's (normed) output feeds the heads, act.py:270
        data_dict["a_hat"] = _lin(self.action_head, hs)
        data_dict["is_pad_hat"] = _lin(self.is_pad_head, hs)
        return data_dict

    def forward_loss(self, data_dict):
        from . import staging

        # produced by the CVAE encoder (the stage under the "transformer.encoder" boundary), consumed here at the top
        mu, logvar = staging.cut("transformer.encoder", data_dict["mu"], data_dict["logvar"], consumed_above="transformer.decoder")
        from . import fused_ops

        if type(self) is ACTPCD and fused_ops.act_loss_supported(data_dict["a_hat"], data_dict["actions"], data_dict["is_pad"], mu, logvar,
                                                                   self.action_loss, self.klloss):
            # one launch each way instead of ~27 (csrc/tokens.hip); same formula, fixed summation order
            data_dict["loss"], data_dict["action_loss"], data_dict["kl_loss"] = fused_ops.act_loss(
                data_dict["a_hat"], data_dict["actions"], data_dict["is_pad"], mu, logvar, self.kl_weight)
            return data_dict
        total_kld = self.klloss(mu, logvar)
        action_loss = self.action_loss(data_dict["a_hat"].float(), data_dict["actions"])
        action_loss = (action_loss * ~data_dict["is_pad"].unsqueeze(-1)).mean()
        data_dict["action_loss"] = action_loss
        data_dict["kl_loss"] = total_kld
        data_dict["loss"] = action_loss + total_kld * self.kl_weight
        return data_dict

    def backward_stages(self):
        """Parameters grouped by WHEN their gradient is complete in backward (latest-used first), each with the name of the
        staging.cut that bounds the stage from below (None: runs to the inputs).  Used by the trainer to exchange a
        stage's gradients while the next stage still computes."""
        dec = [self.query_embed.weight] + list(self.action_head.parameters()) + list(self.is_pad_head.parameters()) + \
            list(self.transformer.decoder.parameters())
        enc = list(self.transformer.encoder.parameters())
        tok = self.tokenizer_parameters()
        seen = {id(p) for p in dec + enc + tok}
        rest = [p for p in self.parameters() if id(p) not in seen]  # CVAE encoder, input / latent projections, embeddings
        return [("transformer.decoder", dec), ("transformer.encoder", enc), ("tokens", rest), (None, tok)]

    def fused_batchnorms(self):
        return [self.bn] if self.sa_impl == "fused" else []

    def tokenizer_parameters(self):
        """Parameters used by `forward(..., stage="tokenize")` (PointNet + the SA layer): everything whose shapes follow
        the number of points; the rest of the policy sees only the fixed-size token matrix."""
        return list(self.backbone.parameters()) + list(self.linear.parameters()) + list(self.bn.parameters())

    @staticmethod
    def hybrid_split(batch):
        """-> (the ragged part handed to stage "tokenize", the fixed-shape rest)."""
        return {"pcds": batch["pcds"]}, {k: v for k, v in batch.items() if k != "pcds"}

    @staticmethod
    def hybrid_merge(rest, boundary):
        return dict(rest, pcd_embed=boundary)

    def forward(self, data_dict, stage=None):
        if stage == "tokenize":  # point clouds -> (tokens (B, H, 1, M), position embedding): the ragged half of the step
            return self.forward_pcd_embed(data_dict["pcds"])
        # The CVAE encoder (102 tokens) and the point-cloud tokenizer are independent until the decoder:
        # on the GPU the former runs on a forked HIP stream (its backward follows on the same stream), so
        # its ~200 small kernels fill the gaps of the PointNet / set-abstraction branch.
        fork = self.overlap_sampling and getattr(self, "fork_cvae", True) and data_dict["qpos"].is_cuda
        if fork:
            from .. import _graphs

            # a step captured as a chain of graphs (data parallel, _graphs.SegmentedCapture) is cut at the tokenizer's BatchNorm
            # collectives: a capture cannot be ended while a forked stream is still unjoined, so the branch runs in line there.
            # (Measured and dropped: the branch as a graph of its own, replayed on a second stream beside the tokenizer's segments
            # and joined in front of the decoder -- 5.55 ms against 5.34 ms in line at C2, 7.29 against 7.08 at C4: two more cuts
            # and a cross-stream fork / join per step cost more than the 0.3 ms of overlap they buy.)
            fork = not _graphs.cuts()
        if fork:
            main = torch.cuda.current_stream(data_dict["qpos"].device)
            side = self.__dict__.get("_cvae_stream")
            if side is None:
                side = self.__dict__["_cvae_stream"] = torch.cuda.Stream(device=data_dict["qpos"].device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                data_dict = self.forward_encoder(data_dict)
            latent = data_dict["latent_input"]
            data_dict = self.forward_obs_embed(data_dict)
            main.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                for t in (latent, data_dict["mu"], data_dict["logvar"]):
                    if t is not None:
                        t.record_stream(main)
        else:
            data_dict = self.forward_encoder(data_dict)
            data_dict = self.forward_obs_embed(data_dict)
        data_dict = self.forward_decoder(data_dict)
        if not data_dict["is_training"]:
            return data_dict
        return self.forward_loss(data_dict)


class ACTRLBenchPCD(ACTPCD):
    """RLBench variant of the point-cloud ACT (/root/reference/src/models/components/act/act.py:707-825): the action is
    (position 3, rotation in the 6-D representation, gripper open, [collision flag]); gripper / collision go through a
    sigmoid, the position error is weighted, and at rollout time the rotation leaves as a quaternion.  Same constructor
    arguments and state-dict keys as the reference class (configs/model/rlbench_act_pcd_model.yaml:20-65)."""

    def __init__(self, backbone, transformer, encoder, hidden_dim, num_queries, num_cameras=0, action_dim=8, qpos_dim=9,
                 env_state_dim=0, latent_dim=32, action_loss=None, klloss=None, kl_weight=20.0, goal_cond_dim=0,
                 obs_feature_pos_embedding=None, freeze_backbone=False, pcd_nsample=16, pcd_npoints=1024, sampling="fps",
                 heatmap_th=0.1, ignore_vae=False, rot_type="6d", collision=False, position_loss_weight=1.0, use_mask=False,
                 bg_ratio=0.0, **kwargs):
        super().__init__(backbone=backbone, transformer=transformer, encoder=encoder, hidden_dim=hidden_dim,
                         num_queries=num_queries, num_cameras=0, action_dim=action_dim, qpos_dim=qpos_dim,
                         env_state_dim=env_state_dim, latent_dim=latent_dim, action_loss=action_loss, klloss=klloss,
                         kl_weight=kl_weight, goal_cond_dim=goal_cond_dim, obs_feature_pos_embedding=None,
                         freeze_backbone=freeze_backbone, pcd_nsample=pcd_nsample, pcd_npoints=pcd_npoints, sampling=sampling,
                         heatmap_th=heatmap_th, ignore_vae=ignore_vae, use_mask=use_mask, bg_ratio=bg_ratio, **kwargs)
        if rot_type != "6d":
            raise NotImplementedError(rot_type)  # the reference's rollout branch raises for anything else (act.py:790-794)
        self.rot_type = rot_type
        self.collision = collision
        self.position_loss_weight = position_loss_weight

    def forward_decoder(self, data_dict):
        hs = self.transformer(
            data_dict["src"], None, self.query_embed.weight, data_dict["pos"], data_dict["latent_input"],
            data_dict["proprio_input"], self.additional_pos_embed.weight,
        )[0]
        a_hat = _lin(self.action_head, hs)  # (B, num_queries, action_dim)
        position = a_hat[..., :3]
        if self.collision:  # (..., gripper, collision) are the last two entries
            gripper = torch.sigmoid(a_hat[..., -2:])
            rot = a_hat[..., 3:-2]
        else:
            gripper = torch.sigmoid(a_hat[..., -1:])
            rot = a_hat[..., 3:-1]
        if not data_dict["is_training"]:
            from .rotations import matrix_to_quaternion, rotation_6d_to_matrix

            rot = matrix_to_quaternion(rotation_6d_to_matrix(rot.float()))
        data_dict["a_hat"] = torch.cat([position, rot, gripper], dim=-1)
        data_dict["is_pad_hat"] = _lin(self.is_pad_head, hs)
        return data_dict

    def forward_loss(self, data_dict):
        from . import staging

        mu, logvar = staging.cut("transformer.encoder", data_dict["mu"], data_dict["logvar"], consumed_above="transformer.decoder")
        total_kld = self.klloss(mu, logvar)
        action_loss = self.action_loss(data_dict["a_hat"].float(), data_dict["actions"])
        weight = action_loss.new_ones(action_loss.shape[-1])
        weight[:3] = self.position_loss_weight  # act.py:816: the position error counts position_loss_weight times
        action_loss = (action_loss * weight * ~data_dict["is_pad"].unsqueeze(-1)).mean()
        data_dict["action_loss"] = action_loss
        data_dict["kl_loss"] = total_kld
        data_dict["loss"] = action_loss + total_kld * self.kl_weight
        return data_dict
