"""polus/ner/models.py drop-in: the NER head over 768-d BERT embeddings,
Dense(768->128, swish) -> Dense(128->C) -> CRF (:26-44) and its dropout variant (:46-66)."""
import torch

from .. import ops
from ..layers import CRF, Dense, Dropout
from ..models import Sequential, SavableModel, from_config, resolve_activation


class NERBertModel(SavableModel):
    def inference(self, x):
        """polus/ner/models.py:12-15: argmax over the (one-hot Viterbi) output, int32."""
        out = self(x, training=False)
        o2 = out.reshape(-1, out.shape[-1]).float().contiguous()
        res = torch.empty(o2.shape[0], dtype=torch.int32, device=o2.device)
        ops.argmax(o2, res)
        return res.view(out.shape[:-1])


class SequentialNERBertModel(Sequential, NERBertModel):
    pass


@from_config
def baselineNER_MLP_CRF(sequence_length=256, output_classes=3, hidden_space=128, activation="swish",
                        compute_dtype="f32", input_dim=768, **kwargs):
    crf_layer = CRF(output_classes)
    model = SequentialNERBertModel([
        Dense(hidden_space, activation=activation, input_shape=(sequence_length, input_dim)),
        Dense(output_classes, out_dtype=torch.float32),
        crf_layer,
    ], compute_dtype=compute_dtype, input_dim=input_dim, name=kwargs.get("name", "baselineNER_MLP_CRF"))
    model.loss = crf_layer.loss
    model.loss_sample_weights = crf_layer.loss_sample_weights
    return model


@from_config
def baselineNER_MLP_Dropout_CRF(sequence_length=256, output_classes=3, hidden_space=128, droupout_p=0.0,
                                activation="swish", compute_dtype="f32", input_dim=768, **kwargs):
    crf_layer = CRF(output_classes)
    model = SequentialNERBertModel([
        Dropout(droupout_p, input_shape=(sequence_length, input_dim)),
        Dense(hidden_space, activation=activation),
        Dense(output_classes, out_dtype=torch.float32),
        crf_layer,
    ], compute_dtype=compute_dtype, input_dim=input_dim, name=kwargs.get("name", "baselineNER_MLP_Dropout_CRF"))
    model.loss = crf_layer.loss
    model.loss_sample_weights = crf_layer.loss_sample_weights
    return model
