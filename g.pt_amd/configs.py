"""Model shapes of the reference's shipped configurations (configs/train/mnist_loss.yaml:34-47,
configs/train/cifar10_error.yaml:34-47), reduced-width variants with the same token plans for tests, and the
synthetic-input distributions of SURVEY.md section 8(d).  bench.py, the training-step driver and tests/cases.py
all read the shapes from here."""

MNIST_SIZES = [7840, 10, 100, 10]
MNIST_NAMES = ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
CIFAR_SIZES = [432, 4608, 320, 10]
CIFAR_NAMES = ["conv1.weight", "conv2.weight", "fc.weight", "fc.bias"]

# Cartpole policy (data_gen/train_rl.py:95-121 with data_gen/configs/rl/train/Cartpole.yaml:5: pi_hid_sizes [32, 32]; obs 4, act 1;
# Gpt/tasks.py:75): state_dict order = log_std, then the actor's Linears
CARTPOLE_SIZES = [1, 128, 32, 1024, 32, 32, 1]
CARTPOLE_NAMES = ["log_std", "actor.0.weight", "actor.0.bias", "actor.2.weight", "actor.2.bias", "actor.4.weight", "actor.4.bias"]

CASES = {
    # reduced-width models with the real token plans (26 tokens) and the real head dims (96 / 128)
    "mnist_small": dict(parameter_sizes=MNIST_SIZES, parameter_names=MNIST_NAMES, n_embd=192, n_head=2,
                        n_layer=2, chunk_size=1000, split_policy="chunk_within_layer", loss_lo=0.3, loss_hi=2.4,
                        loss_target=0.0, w_std=0.538),
    "cifar_small": dict(parameter_sizes=CIFAR_SIZES, parameter_names=CIFAR_NAMES, n_embd=256, n_head=2,
                        n_layer=2, chunk_size=576, split_policy="chunk_within_layer", loss_lo=40.0, loss_hi=90.0,
                        loss_target=35.0, w_std=0.538),
    "cartpole_small": dict(parameter_sizes=CARTPOLE_SIZES, parameter_names=CARTPOLE_NAMES, n_embd=256, n_head=2,
                           n_layer=2, chunk_size=2048, split_policy="layer_by_layer", loss_lo=10.0, loss_hi=480.0,
                           loss_target=500.0, w_std=0.538),
    # the pretrained configurations (configs/train/mnist_loss.yaml:34-47, cifar10_error.yaml:34-47)
    "mnist": dict(parameter_sizes=MNIST_SIZES, parameter_names=MNIST_NAMES, n_embd=1536, n_head=16,
                  n_layer=12, chunk_size=1000, split_policy="chunk_within_layer", loss_lo=0.3, loss_hi=2.4,
                  loss_target=0.0, w_std=0.538),
    "cifar": dict(parameter_sizes=CIFAR_SIZES, parameter_names=CIFAR_NAMES, n_embd=2048, n_head=16,
                  n_layer=12, chunk_size=576, split_policy="chunk_within_layer", loss_lo=40.0, loss_hi=90.0,
                  loss_target=35.0, w_std=0.538),
    # the third shipped task (configs/train/cartpole.yaml:38-46): one token per parameter tensor, 18 tokens, 12 heads of 128;
    # prompts are returns (Gpt/tasks.py:78-80: maximise, best_prompt 500)
    "cartpole": dict(parameter_sizes=CARTPOLE_SIZES, parameter_names=CARTPOLE_NAMES, n_embd=1536, n_head=12,
                     n_layer=12, chunk_size=2048, split_policy="layer_by_layer", loss_lo=10.0, loss_hi=480.0,
                     loss_target=500.0, w_std=0.538),
    # BASELINE.json configs[4]: CIFAR shapes with a scaled-up transformer (the reference defines none)
    "cifar_big": dict(parameter_sizes=CIFAR_SIZES, parameter_names=CIFAR_NAMES, n_embd=3072, n_head=24,
                      n_layer=24, chunk_size=576, split_policy="chunk_within_layer", loss_lo=40.0, loss_hi=90.0,
                      loss_target=35.0, w_std=0.538),
}


def gpt_kwargs(case):
    c = CASES[case]
    return dict(parameter_sizes=list(c["parameter_sizes"]), parameter_names=list(c["parameter_names"]),
                num_frequencies=128, max_freq_log2=14, predict_xstart=True, absolute_loss_conditioning=True,
                n_embd=c["n_embd"], n_head=c["n_head"], n_layer=c["n_layer"], chunk_size=c["chunk_size"],
                split_policy=c["split_policy"], encoder_depth=1, decoder_depth=1,
                attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0)
