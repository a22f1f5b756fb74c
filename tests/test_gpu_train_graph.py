"""Trainer(graph=True): the training step recorded once into a HIP graph and replayed must do what the eager step does -
same losses, same parameters after several steps with changing inputs, new dropout masks on every replay, AdamW's bias
correction advancing."""
import pytest
import torch

pytestmark = pytest.mark.gpu
D = "cuda:0"


def _run(graph, steps, dropout=0.3):
    from image_captioning_amd import transformer_word_level_bahdnau as my_bah
    from image_captioning_amd._bah_train import DropoutState
    from image_captioning_amd.synthetic import seeded_state_dict, synthetic_captions, synthetic_images
    from image_captioning_amd.training import Trainer
    V, B, T = 300, 4, 6
    batches = [(synthetic_images(B, seed=10 + i).to(D), synthetic_captions(B, T, V, 2, 3, 0, seed=20 + i).to(D)) for i in range(3)]
    model = my_bah.build_model(V, dropout=dropout)
    model.load_state_dict(seeded_state_dict(model, seed=5))
    model.to(D)
    DropoutState.manual_seed(123)
    trainer = Trainer(model, lr=1e-3, weight_decay=1e-2, graph=graph)
    losses = []
    for i in range(steps):
        im, cp = batches[i % 3]
        losses.append(trainer.step(im, cp, 0).item())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return losses, sd, trainer, DropoutState.offset


def test_recorded_step_matches_eager_step():
    steps = 7                                        # 2 eager warm-up steps, the recording, 5 replays
    l_e, sd_e, tr_e, off_e = _run(False, steps)
    l_g, sd_g, tr_g, off_g = _run(True, steps)
    assert tr_g._graph is not None
    assert off_e == off_g                            # the host's view of the dropout generator stays in step
    assert tr_e.opt.steps == tr_g.opt.steps == steps and tr_e.opt.param_steps == tr_g.opt.param_steps
    for a, b in zip(l_e, l_g):
        assert abs(a - b) <= 1e-4 * abs(a), (l_e, l_g)
    assert len(set(round(v, 6) for v in l_g)) == steps          # every step saw its own inputs / masks / weights
    bad = []
    for k, want in sd_e.items():
        got = sd_g[k]
        if not want.is_floating_point():
            assert torch.equal(got, want), k                     # num_batches_tracked
            continue
        err = (got - want).abs().max().item()
        if err > 1e-4 * want.abs().max().item() + 1e-6:
            bad.append((k, err, want.abs().max().item()))
    assert not bad, bad[:10]


def test_replays_draw_new_dropout_masks():
    """Same inputs every step, learning rate 0: only the dropout masks can change the loss from replay to replay."""
    from image_captioning_amd import transformer_word_level_bahdnau as my_bah
    from image_captioning_amd._bah_train import DropoutState
    from image_captioning_amd.synthetic import seeded_state_dict, synthetic_captions, synthetic_images
    from image_captioning_amd.training import Trainer
    V, B, T = 300, 4, 6
    im, cp = synthetic_images(B, seed=1).to(D), synthetic_captions(B, T, V, 2, 3, 0, seed=2).to(D)
    model = my_bah.build_model(V, dropout=0.5)
    model.load_state_dict(seeded_state_dict(model, seed=5))
    model.to(D)
    DropoutState.manual_seed(7)
    trainer = Trainer(model, lr=0.0, weight_decay=0.0, graph=True)
    losses = [trainer.step(im, cp, 0).item() for _ in range(6)]
    assert len(set(losses[2:])) == 4, losses


def test_graph_rejects_other_shapes():
    l, sd, trainer, _ = _run(True, 3)
    from image_captioning_amd.synthetic import synthetic_captions, synthetic_images
    with pytest.raises(ValueError):
        trainer.step(synthetic_images(2, seed=1).to(D), synthetic_captions(2, 6, 300, 2, 3, 0, seed=2).to(D), 0)


def test_two_graph_form_with_a_process_group():
    """With a process group up the step is recorded as two graphs with the gradient exchange between them; a single-rank
    group exercises that structure (the exchange itself is covered by the gloo tests on CPU)."""
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29531", rank=0, world_size=1)
    try:
        l_g, sd_g, tr_g, _ = _run(True, 6)
        assert tr_g.reducer is not None and tr_g._graph_opt is not None
    finally:
        dist.destroy_process_group()
    l_e, sd_e, _, _ = _run(False, 6)
    for a, b in zip(l_e, l_g):
        assert abs(a - b) <= 1e-4 * abs(a), (l_e, l_g)
    for k, want in sd_e.items():
        if want.is_floating_point():
            assert (sd_g[k] - want).abs().max().item() <= 1e-4 * want.abs().max().item() + 1e-6, k


@pytest.mark.parametrize("kind", ["word_transformer", "seq2seq_lstm"])
def test_recorded_step_for_the_other_models(kind):
    """The word-level transformer (compute_loss) and the ResNet-18 + LSTM model (CrossEntropyLoss, frozen trunk with
    train-mode BatchNorm): recorded steps against eager steps."""
    from image_captioning_amd._bah_train import DropoutState
    from image_captioning_amd.synthetic import seeded_state_dict, synthetic_captions, synthetic_images
    from image_captioning_amd.training import Trainer
    V, B, T = 300, 4, 6
    batches = [(synthetic_images(B, seed=30 + i).to(D), synthetic_captions(B, T, V, 2, 3, 0, seed=40 + i).to(D)) for i in range(2)]

    def run(graph):
        if kind == "word_transformer":
            from image_captioning_amd import transformer_word_level as twl
            model, loss = twl.build_model(V), "compute_loss"
        else:
            from image_captioning_amd import seq2seqbasic as s2s
            model, loss = s2s.ImageCaptioningModel(256, 512, V, "resnet18", "lstm", 2), "cross_entropy"
        model.load_state_dict(seeded_state_dict(model, seed=6))
        model.to(D)
        DropoutState.manual_seed(5)
        trainer = Trainer(model, lr=1e-3, weight_decay=1e-2, loss=loss, graph=graph)
        losses = [trainer.step(*batches[i % 2], 0).item() for i in range(6)]
        return losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, trainer

    l_e, sd_e, _ = run(False)
    l_g, sd_g, tr = run(True)
    assert tr._graph is not None
    for a, b in zip(l_e, l_g):
        assert abs(a - b) <= 1e-4 * abs(a), (l_e, l_g)
    for k, want in sd_e.items():
        if want.is_floating_point():
            assert (sd_g[k] - want).abs().max().item() <= 1e-4 * want.abs().max().item() + 1e-6, k
        else:
            assert torch.equal(sd_g[k], want), k


def test_eager_steps_release_their_activations_without_the_cycle_collector():
    """A step's saved activations must go when its autograd graph goes (reference counts), not when Python's cycle
    collector next runs: with the collector switched off, device memory after the fifth step equals that after the
    second."""
    import gc
    from image_captioning_amd import transformer_word_level_bahdnau as my_bah
    from image_captioning_amd.synthetic import seeded_state_dict, synthetic_captions, synthetic_images
    from image_captioning_amd.training import Trainer
    V, B, T = 300, 8, 6
    im, cp = synthetic_images(B, seed=1).to(D), synthetic_captions(B, T, V, 2, 3, 0, seed=2).to(D)
    model = my_bah.build_model(V, dropout=0.3)
    model.load_state_dict(seeded_state_dict(model, seed=5))
    model.to(D)
    trainer = Trainer(model, lr=1e-3, weight_decay=1e-2)
    gc.collect()
    gc.disable()
    try:
        seen = []
        for _ in range(5):
            trainer.step(im, cp, 0)
            torch.cuda.synchronize()
            seen.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    assert seen[4] <= seen[1] + (1 << 20), seen
