"""`ASR`: the recogniser object of the reference's test_asr.py (:14-220), on the MI355X models of this package.

    from tensorflowasr_amd.config import UserConfig
    from tensorflowasr_amd.asr import ASR
    asr = ASR(UserConfig('configs/am_data.yml', 'configs/conformerS.yml'))
    phones, text = asr.stt('utt.wav')

Same construction (featurizers from `inp_config` / `tar_config` / `speech_config`, encoder / CTCDecoder / Translator
from `model_config`), same checkpoint directory convention (`<outdir>/{encoder,ctc_decoder,translator}-ckpt/
model_<step>.<ext>`, highest step wins; the reference's Keras `.h5` files, TensorFlow checkpoints and `.npz` files of
Keras-layout tensors are all read directly -- checkpoint.latest_checkpoint), same
outputs (`' '.join(phones), ''.join(text)`).  Everything between the waveform and the token ids runs through
libmi355asr.so; there is no CPU path."""
import logging
import os
import time

import numpy as np

from .featurizers import SpeechFeaturizer, TextFeaturizer
from .models import ConformerEncoder, CTCDecoder, StreamingConformerEncoder, Translator, ctc_greedy_decode


class ASR:
    def __init__(self, config, device="cuda:0", load_checkpoint=True, verbose=False):
        self.running_config = config["running_config"]
        self.speech_config = config["speech_config"]
        self.model_config = config["model_config"]
        self.opt_config = config["optimizer_config"] if "optimizer_config" in config else None
        self.phone_featurizer = TextFeaturizer(config["inp_config"])
        self.text_featurizer = TextFeaturizer(config["tar_config"])
        self.speech_featurizer = SpeechFeaturizer(self.speech_config)
        self.chunk = self.speech_config["sample_rate"] * self.speech_config["streaming_bucket"]
        self.device = device
        self.verbose = verbose
        self.timings = {}
        self.compile(load_checkpoint)

    # test_asr.py:26-93
    def compile(self, load_checkpoint=True):
        mc, sc = self.model_config, self.speech_config
        enc_kw = dict(dmodel=mc["dmodel"], reduction_factor=mc["reduction_factor"], num_blocks=mc["num_blocks"],
                      head_size=mc["head_size"], num_heads=mc["num_heads"], kernel_size=mc["kernel_size"],
                      fc_factor=mc["fc_factor"], dropout=mc["dropout"], add_wav_info=sc["add_wav_info"],
                      sample_rate=sc["sample_rate"], n_mels=sc["num_feature_bins"],
                      mel_layer_type=sc["mel_layer_type"], mel_layer_trainable=sc["mel_layer_trainable"],
                      stride_ms=sc["stride_ms"], device=self.device)
        if not sc["streaming"]:
            self.encoder = ConformerEncoder(name="conformer_encoder", **enc_kw)
        else:
            assert "Streaming" in mc["name"], "am_data.yml set streaming=True,But model.yml is OfflineCTC"
            self.encoder = StreamingConformerEncoder(name="stream_conformer_encoder", **enc_kw)
            self.encoder.add_chunk_size(
                chunk_size=int(sc["streaming_bucket"] * sc["sample_rate"]), mel_size=sc["num_feature_bins"],
                hop_size=int(sc["stride_ms"] * sc["sample_rate"] // 1000) * mc["reduction_factor"])
            self.encoder.set_inference_func()
        self.ctc_model = CTCDecoder(num_classes=self.phone_featurizer.num_classes, dmodel=mc["dmodel"],
                                    num_blocks=mc["ctcdecoder_num_blocks"], head_size=mc["head_size"],
                                    num_heads=mc["num_heads"], kernel_size=mc["ctcdecoder_kernel_size"],
                                    dropout=mc["ctcdecoder_dropout"], fc_factor=mc["ctcdecoder_fc_factor"],
                                    device=self.device)
        self.translator = Translator(inp_classes=self.phone_featurizer.num_classes,
                                     tar_classes=self.text_featurizer.num_classes, dmodel=mc["dmodel"],
                                     num_blocks=mc["translator_num_blocks"], head_size=mc["head_size"],
                                     num_heads=mc["num_heads"], kernel_size=mc["translator_kernel_size"],
                                     dropout=mc["translator_dropout"], fc_factor=mc["translator_fc_factor"],
                                     device=self.device)
        self.encoder._build()
        self.ctc_model._build()
        self.translator._build()
        self.translator.set_inference_func()
        if load_checkpoint:
            self.load_checkpoint()
        if self.verbose:
            self.encoder.summary(line_length=100)
            self.ctc_model.summary(line_length=100)
            self.translator.summary(line_length=100)

    # test_asr.py:95-114
    @staticmethod
    def _latest(checkpoint_dir):
        from .checkpoint import latest_checkpoint
        return latest_checkpoint(checkpoint_dir)

    def load_checkpoint(self):
        for sub, model, by_name in (("encoder-ckpt", self.encoder, True), ("ctc_decoder-ckpt", self.ctc_model, False),
                                    ("translator-ckpt", self.translator, False)):
            path = self._latest(os.path.join(self.running_config["outdir"], sub))
            model.load_weights(path, by_name=by_name)
            logging.info("%s load at %s", sub.replace("-ckpt", ""), path)

    # ---- decoding helpers --------------------------------------------------------------------------------
    def _phone_ids(self, enc_outputs):
        """softmax + tf.keras.backend.ctc_decode(greedy) + clip(-1 -> 0) (test_asr.py:196-200): per-frame argmax
        inside the CTC head kernel, merge/blank-drop on the device."""
        _, frame_ids = self.ctc_model(enc_outputs, training=False, return_argmax=True, return_logits=False)
        # tf.keras.backend.ctc_decode: the blank is the LAST class (num_classes - 1), independent of `blank_at_zero`
        ids, lens = ctc_greedy_decode(frame_ids, None, blank=self.phone_featurizer.num_classes - 1)
        # ctc_decode's dense output is as wide as the longest decoded sequence of the batch, padded with -1; the
        # width matters: the Translator has no mask, padded positions reach their neighbours through its ConvModule
        width = int(lens.max().item())
        return ids[:, :width].clamp_(min=0).contiguous(), lens

    def _finish(self, ctc_decode_row, translator_row):
        ctc_result = [int(n) for n in ctc_decode_row if n != 0]
        txt_result = []
        for n in translator_row:
            n = int(n)
            if n != 0:
                txt_result.append(n)
            if n == self.text_featurizer.endid():
                break
        phone = self.phone_featurizer.iextract(ctc_result)
        txt = self.text_featurizer.iextract(txt_result)
        return " ".join(phone), "".join(txt)

    # test_asr.py:186-219
    def offline_stt(self, wav_path):
        data = self.speech_featurizer.load_wav(wav_path)
        input_wav = data.reshape([1, -1, 1])
        t0 = time.time()
        enc_outputs = self.encoder(input_wav, training=False)
        ctc_decode, _ = self._phone_ids(enc_outputs)
        _, translator_out = self.translator([ctc_decode, enc_outputs], training=False, return_argmax=True, return_logits=False)
        ctc_row, txt_row = ctc_decode[0].cpu().numpy(), translator_out[0].cpu().numpy()
        self.timings["offline_stt"] = time.time() - t0
        return self._finish(ctc_row, txt_row)

    # test_asr.py:116-164
    def stream_stt(self, wav_path):
        import torch
        data = self.speech_featurizer.load_wav(wav_path)
        enc_outputs, ctc_row, txt_row = None, [], []
        for i in range(9999):
            s = i * self.chunk
            e = s + self.chunk
            if s >= len(data):
                break
            input_wav = data[int(s):int(e)]
            if len(input_wav) < int(self.chunk):      # the reference's tf.function pads nothing and would fail on a
                input_wav = np.pad(input_wav, (0, int(self.chunk) - len(input_wav)))   # short tail: zero-pad it
            enc_output = self.encoder.inference(input_wav.reshape([1, -1, 1]))
            enc_outputs = enc_output if enc_outputs is None else torch.cat((enc_outputs, enc_output), 1)
            ctc_decode, _ = self._phone_ids(enc_outputs)           # global CTC over everything heard so far
            ctc_row = ctc_decode[0].cpu().numpy()
            ctc_result = [int(n) for n in ctc_row if n != 0] + [0] * 10
            _, tr = self.translator([np.array([ctc_result], "int32"), enc_outputs], return_argmax=True, return_logits=False)
            txt_row = tr[0].cpu().numpy()
        return self._finish(ctc_row, txt_row)

    def stt(self, wav_path):
        if self.speech_config["streaming"]:
            return self.stream_stt(wav_path)
        return self.offline_stt(wav_path)

    # test_asr.py:166-185 (host-side equivalents the reference keeps next to the TF path)
    @staticmethod
    def remove_blank(labels, blank=0):
        new_labels, previous = [], None
        for l in labels:
            if l != previous:
                new_labels.append(l)
                previous = l
        return [l for l in new_labels if l != blank]

    def greedy_decode(self, y, blank=1331):
        return self.remove_blank(np.argmax(y, axis=1), blank)
