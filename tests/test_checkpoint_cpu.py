"""CPU: local HF checkpoint import — name mapping and Q/K/V fusion checked by running the oracle on
the imported weights against the PyTorch BertModel that wrote the file (random init, no download)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import bert as ob
from polus_amd.checkpoint import read_local_hf_checkpoint


def test_local_safetensors_import_matches_hf(tmp_path):
    from safetensors.numpy import save_file
    from transformers import BertConfig, BertModel
    hc = BertConfig(vocab_size=60, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                    max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, pad_token_id=None)
    hc._attn_implementation = "eager"
    torch.manual_seed(3)
    m = BertModel(hc, add_pooling_layer=False).eval()
    sd = {"bert." + k: v.detach().numpy().copy() for k, v in m.state_dict().items() if "position_ids" not in k}
    save_file(sd, str(tmp_path / "model.safetensors"))
    json.dump(hc.to_dict(), open(tmp_path / "config.json", "w"), default=str)
    cfg_d, params = read_local_hf_checkpoint(str(tmp_path))
    assert cfg_d["hidden_size"] == 128 and params["layer1.qkv.w"].shape == (384, 128)
    ocfg = ob.BertConfig(60, 128, 2, 2, 256, 32, 2)
    ids = np.random.default_rng(0).integers(0, 60, size=(2, 10)).astype(np.int32)
    mask = np.ones((2, 10), np.int32); mask[1, 7:] = 0
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    last, pooled, _ = ob.bert_fwd(p64, ocfg, ids, mask)
    with torch.no_grad():
        emb = m.embeddings(input_ids=torch.tensor(ids, dtype=torch.long))
        add = (1.0 - torch.tensor(mask, dtype=torch.float32))[:, None, None, :] * -10000.0
        ref = m.encoder(emb, attention_mask=add).last_hidden_state.numpy()
    assert np.abs(last - ref).max() < 2e-5
    assert np.array_equal(pooled, last[:, 0, :])


def test_by_name_checkpoints_are_refused():
    from polus_amd.checkpoint import load_bert_from_local
    with pytest.raises(FileNotFoundError):
        load_bert_from_local("bert-base-uncased")
