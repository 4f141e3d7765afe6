"""fp32 CPU restatement of the reference's representation / dynamics / prediction forward.

TEST INFRASTRUCTURE ONLY.  The reference runs these blocks through DJL on the PyTorch native
engine, i.e. the very ATen ops called here (conv2d, layer_norm, linear, softmax), so this is the
fp32 reference the HIP forward is compared against (tolerance stated in the tests; the reference's
own cross-backend precedent is allClose(1e-2, 1e-2), onnx/.../BlockTestHelper.java:36).
No stored activations exist in the reference => network parity is "unpinned" beyond this
restatement.

Citations (platform/src/main/java/ai/enpasos/muzero/platform/agent/d_model/djl/blocks/ = B/,
onnx/onnxModelGen/src/main/java/ai/enpasos/mnist/blocks/ = X/):
  B/c_mainfunctions/RepresentationBlock.java:38-40, DynamicsBlock.java:40-42,
  MainRepresentationOrDynamicsBlock.java:42-54, PredictionBlock.java:46-84,
  B/d_lowerlevel/StartResidualBlock.java:53-62, ConcatInputsBlock.java:54-60, ResidualTower.java:30-40,
  BottleneckResidualBlock.java:55-79, BroadcastResidualBlock.java:56-77, Conv3x3.java:34-43, Conv1x1.java:34-43,
  X/BroadcastBlock.java:39-58, X/ext/RescaleBlockExt.java:49-73, X/ext/LayerNormExt.java:22-27,
  B/b_inference/InitialInferenceBlock.java:66-70, RecurrentInferenceBlock.java:67-71,
  InitialInferenceListTranslator.java:44-112 (softmax, value scaling).
"""
import numpy as np
import torch
import torch.nn.functional as F


class NetOracle:
    def __init__(self, spec, weights, dtype=torch.float32):
        self.s = spec
        self.dtype = dtype
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in weights.items()}

    # --- building blocks ---------------------------------------------------------------------
    def _ln(self, x, name):
        s = self.s
        shp = (x.shape[1], s.height, s.width)
        return F.layer_norm(x, shp, self.w[name + ".g"].reshape(shp), self.w[name + ".b"].reshape(shp), 1e-5)

    def _c1(self, x, name):
        w = self.w[name]
        return F.conv2d(x, w.reshape(w.shape[0], w.shape[1], 1, 1))

    def _c3(self, x, name):
        return F.conv2d(x, self.w[name], padding=1)

    def _bottleneck(self, x, p):
        h = self._c1(F.relu(self._ln(x, p + ".ln1")), p + ".c1")
        h = self._c3(F.relu(self._ln(h, p + ".ln2")), p + ".c2")
        h = self._c3(F.relu(self._ln(h, p + ".ln3")), p + ".c3")
        h = self._c1(F.relu(self._ln(h, p + ".ln4")), p + ".c4")
        return h + x

    def _broadcast(self, x, p):
        s = self.s
        h = self._c1(F.relu(self._ln(x, p + ".ln1")), p + ".c1")
        h = F.relu(self._ln(h, p + ".ln2"))
        n, c = h.shape[0], h.shape[1]
        h = F.linear(h.reshape(n * c, s.hw), self.w[p + ".bw"], self.w[p + ".bb"]).reshape(n, c, s.height, s.width)
        h = self._c1(F.relu(self._ln(h, p + ".ln3")), p + ".c2")
        return h + x

    def _tower(self, x, prefix):
        for i in range(self.s.residuals):
            p = f"{prefix}.b{i}"
            x = self._broadcast(x, p) if self.s.is_broadcast(i) else self._bottleneck(x, p)
        return x

    @staticmethod
    def _rescale(x):
        n = x.shape[0]
        flat = x.reshape(n, -1)
        mn = flat.min(dim=1, keepdim=True).values.reshape(n, 1, 1, 1)
        mx = flat.max(dim=1, keepdim=True).values.reshape(n, 1, 1, 1)
        d = torch.clamp(mx - mn, min=1e-5)
        return (x - mn) / d

    # --- h, g, f -------------------------------------------------------------------------------
    def representation(self, obs):
        return self._rescale(self._tower(self._c3(obs, "rep.conv0"), "rep"))

    def dynamics(self, state, action_planes):
        x = self._c3(torch.cat([state, action_planes], dim=1), "dyn.conv0") + state
        return self._rescale(self._tower(x, "dyn"))

    def prediction(self, state):
        s = self.s
        n = state.shape[0]
        p = F.relu(self._ln(self._c1(state, "pred.p.conv"), "pred.p.ln")).reshape(n, -1)
        logits = F.linear(p, self.w["pred.p.fc.w"], self.w["pred.p.fc.b"])
        v = F.relu(self._ln(self._c1(state, "pred.v.conv"), "pred.v.ln")).reshape(n, -1)
        v = F.relu(F.linear(v, self.w["pred.v.fc1.w"], self.w["pred.v.fc1.b"]))
        v = F.linear(v, self.w["pred.v.fc2.w"], self.w["pred.v.fc2.b"])
        if s.two_players:
            v = torch.tanh(v)
        return logits, v.reshape(n)

    def action_planes(self, action, action_cells):
        """one-hot planes [B, a, H, W] from action indices (Network.java:143-152,221-224)."""
        s = self.s
        n = len(action)
        planes = torch.zeros(n, s.action_planes * s.hw, dtype=self.dtype)
        cells = torch.as_tensor(np.asarray(action_cells)[np.asarray(action)], dtype=torch.long)
        rows = torch.nonzero(cells >= 0).reshape(-1)
        planes[rows, cells[rows]] = 1.0
        return planes.reshape(n, s.action_planes, s.height, s.width)

    def _outputs(self, state):
        logits, v = self.prediction(state)
        policy = torch.softmax(logits, dim=1)
        value = self.s.value_scale * v.to(torch.float64)   # double scale * float (translator :94-97)
        return logits, policy, value

    @torch.no_grad()
    def initial_inference(self, obs):
        """obs [B, O, H, W] -> (hidden [B,C,H,W], logits [B,A], policy [B,A], value fp64 [B])."""
        obs = torch.as_tensor(obs, dtype=self.dtype)
        state = self.representation(obs)
        return (state,) + self._outputs(state)

    @torch.no_grad()
    def recurrent_inference(self, state, action, action_cells):
        state = torch.as_tensor(state, dtype=self.dtype)
        nxt = self.dynamics(state, self.action_planes(action, action_cells))
        return (nxt,) + self._outputs(nxt)
