"""Row N1 / f1 on the decode path (north star: "tasks/open_ended_task.py load[s] it as a drop-in"): beam search over
this package's stateful ``Decoder`` against G16 -- the output of the reference's OWN ``BaseTransformer.beam_search`` over
its own ``Decoder`` and ``BeamSearch`` (tests/golden/make_golden.py:g16).

(1) the package's search driver (``openvivqa_amd.beam``) over the package's decoder, fused and reference reorder;
(2) build container only: the reference's UNMODIFIED ``models/base_transformer.py`` and ``models/modules/beam_search.py``
    (tasks/open_ended_task.py:135 -> base_transformer.py:46-54 -> beam_search.py:85-118) over this package's builders,
    with ``models.modules.containers`` resolved to ``openvivqa_amd.modules.containers`` as INTEGRATION.md section A
    prescribes: G16 reproduced, the decoder IS stateful inside ``statefulness(b_s)``, ``states()`` reaches its caches;
(3) the same WITHOUT the containers alias: refused loudly at construction (it used to decode statelessly, silently).

No GPU here: the kernels are the torch-math stand-ins of tests/mock_ops.py; registry lookups, constructors, the state
machinery, the projected K / V caches and the search control flow are the product's."""
import os
import sys
import types
from types import SimpleNamespace

import pytest
import torch

import mock_ops
from golden_cases import GenVocab, load_case
from openvivqa_amd.config import ConfigNode

REF = "/root/reference"
needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


@pytest.fixture()
def cpu_ops(monkeypatch):
    import openvivqa_amd as A
    import openvivqa_amd.functional as Fn
    import openvivqa_amd.runtime as rt
    import openvivqa_amd.modules.embeddings as emb
    for mod in (A, Fn, rt, emb):  # (A.ops: what the decode helpers' function-level ``from .. import ops`` resolves to)
        monkeypatch.setattr(mod, "ops", mock_ops)
    monkeypatch.setattr(rt, "build_arena", lambda module, device=None, compute_dtype=None: rt.ParamArena(
        rt.collect_groups(module), next(module.parameters()).device, compute_dtype or rt.get_compute_dtype()))
    A.set_compute_dtype(torch.float32)
    yield A
    A.set_compute_dtype(torch.bfloat16)


def _g16():
    case = load_case("G16_beam_search")
    return case, GenVocab(case.meta), ConfigNode(case.meta["cfg"])


def _check(case, beam, toks, lp):
    ref_t, ref_lp = case.out[f"beam{beam}_tokens"], case.out[f"beam{beam}_logp"]
    assert torch.equal(toks.reshape(ref_t.shape), ref_t)
    assert float((lp.reshape(ref_lp.shape) - ref_lp).abs().max()) < 2e-5


@pytest.mark.parametrize("beam", [1, 3])
@pytest.mark.parametrize("reorder", ["fused", "reference"])
def test_package_search_over_package_decoder_reproduces_reference_search(cpu_ops, beam, reorder):
    """openvivqa_amd.beam.BeamSearch (two-stage top-k, whole-buffer histories; Module.reorder_states or the reference's
    apply_to_states closure) over openvivqa_amd's Decoder with its projected K / V caches == the reference's search over
    the reference's decoder: same words for every beam, word scores to 2e-5.  ``GraphedBeamSearch`` is the eager driver
    here (no GPU, no graph): it hands the decoder the per-sample encoder features with ``encoder_group=beam``."""
    from openvivqa_amd.beam import BeamSearch, GraphedBeamSearch
    from openvivqa_amd.modules import Decoder
    case, vocab, cfg = _g16()
    dec = Decoder(cfg, vocab)
    dec.load_state_dict(case.w)
    dec.eval()
    enc, mask = case.inputs["enc"], case.inputs["enc_mask"]
    T = vocab.max_answer_length
    if reorder == "fused":
        toks, lp = GraphedBeamSearch(dec, enc.shape[0], T, vocab.bos_idx, vocab.eos_idx, beam, out_size=beam)(
            enc, mask, use_graph=False)
    else:
        held = {}

        def step(t, prev):  # the reference's protocol: per-beam copies of the encoder features from t = 1 on
            if t == 0:
                held["e"], held["m"] = enc, mask
                prev = torch.full((enc.shape[0], 1), vocab.bos_idx, dtype=torch.long)
            elif t == 1:
                held["e"], held["m"] = enc.repeat_interleave(beam, 0), mask.repeat_interleave(beam, 0)
            return dec(prev, held["e"], held["m"])
        with torch.no_grad(), dec.statefulness(enc.shape[0]):
            toks, lp = BeamSearch(dec, step, enc.shape[0], T, vocab.eos_idx, beam, "cpu", reorder="reference").apply(beam)
    _check(case, beam, toks, lp)
    assert not dec._is_stateful and all(layer.self_attn._kv is None for layer in dec.layers)


# ----------------------------------------------------------------------------------------------------------------------
def _reference_tree(monkeypatch, alias_containers):
    """sys.modules / sys.path as INTEGRATION.md section A leaves them: the reference's packages as bare shells (SURVEY
    8c recipe: their fan-out __init__ files do not run), the hot-path builder modules AND -- when asked -- the stateful
    containers resolved to this package.  Returns what has to be undone."""
    import openvivqa_amd.builders as B
    import openvivqa_amd.modules.containers as C
    saved = dict(sys.modules)
    monkeypatch.syspath_prepend(REF)
    for name in ("builders", "models", "models.modules", "data_utils"):
        shell = types.ModuleType(name)
        shell.__path__ = [os.path.join(REF, name.replace(".", "/"))]
        sys.modules[name] = shell
    tc = types.ModuleType("termcolor")
    tc.colored = lambda s, *a, **k: s
    sys.modules["termcolor"] = tc
    for short in ("attention", "encoder", "decoder", "text_embedding", "vision_embedding"):
        sys.modules[f"builders.{short}_builder"] = getattr(B, f"{short}_builder")
    if alias_containers:
        sys.modules["models.modules.containers"] = C
    return saved


def _restore(saved):
    for k in list(sys.modules):
        if k not in saved:
            del sys.modules[k]
    sys.modules.update(saved)


def _generation_model(ref_bt, build_decoder):
    """What every generation model of the reference is (models/vit_mbert_generation.py:16-53 and its siblings): a
    BaseTransformer subclass that owns ``self.decoder = build_decoder(config.DECODER, vocab)`` and an encoder_forward;
    here the encoder is the identity on given features."""
    class Gen(ref_bt.BaseTransformer):
        def __init__(self, cfg, vocab):
            super().__init__(cfg, vocab)
            self.device = torch.device("cpu")
            self.decoder = build_decoder(cfg, vocab)

        def encoder_forward(self, inp):
            return inp["enc"], inp["enc_mask"]
    return Gen


@needs_reference
@pytest.mark.parametrize("beam", [1, 3])
@pytest.mark.parametrize("grad", [False, True])
def test_reference_base_transformer_and_beam_search_drive_package_decoder(cpu_ops, monkeypatch, beam, grad):
    """The reference's unmodified BaseTransformer.beam_search + BeamSearch over this package's Decoder (built by the
    reference's own ``builders.decoder_builder`` name, resolved to this package) reproduce G16.  ``grad=False`` is how
    tasks/open_ended_task.py:134-135 calls it (torch.no_grad: the projected-cache decode path); with autograd on, the
    decoder keeps the reference's raw-input caches."""
    saved = _reference_tree(monkeypatch, alias_containers=True)
    try:
        import models.base_transformer as ref_bt  # noqa: E402  (the reference's file, unmodified)
        import models.modules.beam_search as ref_bs  # noqa: E402
        import builders.decoder_builder as ref_db  # noqa: E402  (= openvivqa_amd.builders.decoder_builder)
        import openvivqa_amd.modules as M
        assert ref_bt.__file__.startswith(REF) and ref_bs.__file__.startswith(REF)
        assert ref_bt.BeamSearch is ref_bs.BeamSearch and ref_bt.Module is M.containers.Module
        case, vocab, cfg = _g16()
        model = _generation_model(ref_bt, ref_db.build_decoder)(cfg, vocab)
        assert isinstance(model.decoder, M.Decoder)
        model.decoder.load_state_dict(case.w)
        model.eval()
        seen = {}
        real_step = model.step

        def spy(t, prev, **kw):  # inside statefulness(b_s): is the decoder stateful, do states() reach its caches?
            out = real_step(t, prev, **kw)
            if t == 2:
                seen["stateful"] = (model.decoder._is_stateful, all(l.self_attn._is_stateful for l in model.decoder.layers))
                states = list(model.states())
                seen["n_states"] = len(states)
                seen["key_rows"] = [tuple(l.self_attn.running_keys.shape) for l in model.decoder.layers]
                seen["own"] = (tuple(model.encoder_features.shape), tuple(model.decoder.running_seq.shape))
            return out
        model.step = spy
        with torch.set_grad_enabled(grad):
            toks, lp = model.beam_search({"enc": case.inputs["enc"], "enc_mask": case.inputs["enc_mask"]},
                                         batch_size=case.inputs["enc"].shape[0], beam_size=beam, out_size=beam)
        _check(case, beam, toks.detach(), lp.detach())
        b_s, D = case.inputs["enc"].shape[0], cfg.D_MODEL
        assert seen["stateful"] == (True, True)
        assert seen["n_states"] == case.meta["n_states"] == 2 + 2 + 2 * cfg.LAYERS  # the model's 2, the decoder's 2, K + V per layer
        assert seen["key_rows"] == [(b_s * beam, 3, D)] * cfg.LAYERS  # three positions cached for every beam
        assert seen["own"] == ((b_s * beam,) + tuple(case.inputs["enc"].shape[1:]), (b_s * beam, 1))
        assert not model.decoder._is_stateful and model.encoder_features is None  # statefulness left
    finally:
        _restore(saved)


@needs_reference
def test_reference_base_transformer_without_containers_alias_is_refused(cpu_ops, monkeypatch):
    """Without the containers alias the reference's BaseTransformer derives from the reference's OWN Module, whose
    ``isinstance(child, Module)`` recursion (containers.py:20-31,50-63) does not see this package's Decoder: round 3
    decoded statelessly and silently there (one token per step, caches never reordered).  Now the adoption itself raises,
    naming the one-line fix."""
    saved = _reference_tree(monkeypatch, alias_containers=False)
    try:
        import models.base_transformer as ref_bt  # noqa: E402
        import builders.decoder_builder as ref_db  # noqa: E402
        import openvivqa_amd.modules.containers as C
        assert ref_bt.Module is not C.Module and ref_bt.Module.__module__ == "models.modules.containers"
        case, vocab, cfg = _g16()
        with pytest.raises(TypeError, match="models.modules.containers"):
            _generation_model(ref_bt, ref_db.build_decoder)(cfg, vocab)
        # what would happen without the guard (the round-3 behaviour), documented: the decoder never becomes stateful
        C.remove_foreign_parent_guard()
        try:
            model = _generation_model(ref_bt, ref_db.build_decoder)(cfg, vocab)
            with model.statefulness(2):
                assert model.decoder._is_stateful is False and len(list(model.states())) == 2
        finally:
            C.install_foreign_parent_guard()
    finally:
        _restore(saved)


@needs_reference
def test_reference_iterative_mcan_from_unmodified_yaml(cpu_ops, monkeypatch):
    """The one shipped model that joins everything on the path: the reference's UNMODIFIED
    ``models/iterative_mcan.py`` -- MCA stack, ``PositionWiseFeedForward`` used directly (:26), the stateful ``Decoder`` under
    ``tasks/open_ended_task.py:128-169`` -- built from the unmodified ``configs/iterative_mcan.yaml`` over this package's
    builders and containers: the teacher-forced forward and ``beam_search(beam_size=3)`` equal the same glue code over the
    oracle's modules (same weights; the oracle side searches with ``oracle_generate``, its own restated search)."""
    import oracle as O
    from openvivqa_amd.config import get_config
    saved = _reference_tree(monkeypatch, alias_containers=True)
    try:
        import models.iterative_mcan as ref_im  # noqa: E402  (the reference's file, unmodified)
        import openvivqa_amd.modules as M
        assert ref_im.__file__.startswith(REF)
        cfg = get_config(os.path.join(REF, "configs", "iterative_mcan.yaml")).MODEL
        cfg.DEVICE = "cpu"

        class Vocab:
            max_answer_length, padding_idx, bos_idx, eos_idx = 7, 0, 1, 2

            def __len__(self):
                return 40
        vocab = Vocab()
        torch.manual_seed(21)
        ours = ref_im.IterativeMCAN(cfg, vocab)
        assert type(ours).__module__ == "models.iterative_mcan" and isinstance(ours.decoder, M.Decoder)
        assert isinstance(ours.self_encoder, M.Encoder) and isinstance(ours.guided_encoder, M.GuidedAttentionEncoder)
        assert type(ours.fusion).__module__ == "models.modules.positionwise_feed_forward"  # the reference's own class (:26)
        text = {"UsualEmbedding": O.OracleUsualEmbedding, "LSTMTextEmbedding": O.OracleLSTMTextEmbedding}
        swaps = dict(build_encoder=O.build_oracle_encoder, build_decoder=lambda c, vocab: O.OracleDecoder(c, vocab),
                     build_text_embedding=lambda c, v: text[c.ARCHITECTURE](c, v),
                     build_vision_embedding=lambda c: O.OracleFeatureEmbedding(c))
        kept = {k: getattr(ref_im, k) for k in swaps}
        try:  # the same glue code over the oracle's modules
            for k, v in swaps.items():
                setattr(ref_im, k, v)
            theirs = ref_im.IterativeMCAN(cfg, vocab)
        finally:
            for k, v in kept.items():
                setattr(ref_im, k, v)
        missing, unexpected = theirs.load_state_dict(ours.state_dict(), strict=False)
        assert not unexpected and not [k for k in missing if "running_" not in k and "pos_emb" not in k], (missing, unexpected)
        with torch.no_grad():  # spread the next-word scores so that no two candidates of the search are close
            for m in (ours, theirs):
                m.decoder.fc.weight.mul_(6.0)
        ours.eval()
        theirs.eval()
        g = torch.Generator().manual_seed(4)
        regions = torch.randn(3, 9, cfg.VISION_EMBEDDING.D_FEATURE, generator=g)
        regions[1, 7:] = 0
        question = torch.randint(4, 40, (3, 6), generator=g)
        question[2, 4:] = 0
        answer = torch.randint(4, 40, (3, 5), generator=g)
        answer[:, 0] = vocab.bos_idx
        answer[0, 3:] = 0
        inp = SimpleNamespace(region_features=regions, question_tokens=question, answer_tokens=answer)
        with torch.no_grad():
            lo, lt = ours(inp), theirs(inp)
        assert lo.shape == (3, 5, 40) and float((lo - lt).abs().max()) < 2e-5
        # decode as tasks/open_ended_task.py:134-135 does: model.beam_search under torch.no_grad()
        seen = {}
        real_step = ours.step

        def spy(t, prev, **kw):
            out = real_step(t, prev, **kw)
            if t == 2:
                seen["stateful"] = ours.decoder._is_stateful
                seen["n_states"] = len(list(ours.states()))
            return out
        ours.step = spy
        with torch.no_grad():
            toks, lp = ours.beam_search(inp, batch_size=3, beam_size=3, out_size=1, return_probs=False)
            enc_t, mask_t = theirs.encoder_forward(inp)
            ref_toks, ref_lp = O.oracle_generate(theirs.decoder, enc_t, mask_t, vocab.bos_idx, vocab.eos_idx, 3,
                                                 max_len=vocab.max_answer_length)
        assert seen["stateful"] is True and seen["n_states"] == 2 + 2 + 2 * cfg.DECODER.LAYERS
        assert torch.equal(toks.reshape(ref_toks.shape), ref_toks)
        assert float((lp.reshape(ref_lp.shape) - ref_lp).abs().max()) < 5e-5
        assert not ours.decoder._is_stateful
    finally:
        _restore(saved)
