"""`ChunkASR` and `ChunkAMTester`: the recogniser of the reference's test_chunk_asr.py (:21-139) and the evaluation loop
of asr/tester/chunk_tester.py (:14-80) for the ChunkConformer, on the MI355X model of this package.

    from tensorflowasr_amd.config import UserConfig
    from tensorflowasr_amd.chunk_asr import ChunkASR
    asr = ChunkASR(UserConfig('configs/am_data.yml', 'configs/chunk_conformerS.yml'))
    for t, phones, text in asr.stream_call('utt.wav')["streaming"]: ...

Same construction as the reference (`ChunkConformer(config, phone_num_classes, text_num_classes)`, weights from the newest
TensorFlow checkpoint under `<outdir>/all-ckpt`, test_chunk_asr.py:36-43), same streaming loop: `wav_buf_length` samples
per call -> picker_stream_predict -> feature_pick -> decoder_stream_predict, greedy CTC decode of [valid | unvalid] text
logits after every call, and the offline `predict` of the whole utterance for comparison.  Every tensor operation between
the waveform and the token ids runs in libmi355asr.so."""
import logging
import os

import numpy as np
import torch

from .eval import _Mean, wer
from .featurizers import SpeechFeaturizer, TextFeaturizer
from .models import ChunkConformer, ctc_greedy_decode, frame_argmax


def _ctc_text(logits, blank):
    """softmax + tf.keras.backend.ctc_decode(greedy) + clip (test_chunk_asr.py:88-100): [1, T, V] logits -> id list without
    zeros.  The argmax of softmax(x) is the argmax of x; the blank is the last class."""
    if logits.shape[1] == 0:
        return []
    ids, lens = ctc_greedy_decode(frame_argmax(logits), None, blank=blank)
    row = ids[0, :int(lens[0].item())].clamp(min=0).cpu().numpy()
    return [int(n) for n in row if n != 0]


class ChunkASR:
    def __init__(self, config, device="cuda:0", load_checkpoint=True):
        self.running_config = config["running_config"]
        self.speech_config = config["speech_config"]
        self.model_config = config["model_config"]
        self.opt_config = config["optimizer_config"] if "optimizer_config" in config else None
        self.phone_featurizer = TextFeaturizer(config["inp_config"])
        self.text_featurizer = TextFeaturizer(config["tar_config"])
        self.speech_featurizer = SpeechFeaturizer(self.speech_config)
        self.config = config
        self.device = device
        self.compile(load_checkpoint)

    # test_chunk_asr.py:35-45
    def compile(self, load_checkpoint=True):
        self.runner = ChunkConformer(self.config, self.phone_featurizer.num_classes, self.text_featurizer.num_classes,
                                     device=self.device)
        self.runner._build()
        if load_checkpoint:
            from .checkpoint import latest_tf_checkpoint
            path = latest_tf_checkpoint(os.path.join(self.running_config["outdir"], "all-ckpt"))
            self.runner.load_weights(path)
            logging.info("ChunkConformer load at %s", path)
        chunk_num, hop, _, _ = self.runner._stream_cfg()
        self.wav_buf_length = chunk_num * hop          # ChunkConformerFront.wav_buf_length (chunk_conformer_blocks.py:419)

    def load_wav(self, wav_path):
        data = self.speech_featurizer.load_wav(wav_path)
        return data / np.abs(data.max())               # test_chunk_asr.py:49 (sic: abs of the max, not max of the abs)

    def offline_stt(self, wav_path):
        """runner.predict on the whole utterance + greedy CTC decode (test_chunk_asr.py:56, 124-137) -> text"""
        data = self.load_wav(wav_path)
        logits, _ = self.runner.predict(data.reshape([1, -1, 1]))
        return "".join(self.text_featurizer.iextract(_ctc_text(logits, self.text_featurizer.num_classes - 1)))

    def stream_call(self, wav_path, verbose=False):
        """test_chunk_asr.py:47-139 -> {"streaming": [(seconds_heard, phones, text), ...], "offline": text}"""
        r = self.runner
        data = self.load_wav(wav_path)
        dev = r._h.device
        caches, caches2 = r.init_picker_caches(1), r.init_decoder_caches(1)
        Vt, Vp = self.text_featurizer.num_classes, self.phone_featurizer.num_classes
        valid_txt_outs = torch.zeros((1, 0, Vt), device=dev)
        valid_phone_outs = torch.zeros((1, 0, Vp), device=dev)
        unvalid_txt_outs = torch.zeros((1, 0, Vt), device=dev)
        out = []
        for i in range(99999):
            s = i * self.wav_buf_length
            e = s + self.wav_buf_length
            if s >= len(data):
                break
            input_wav = data[int(s):int(e)].reshape([1, -1, 1])
            valid_phone_out, _, valid_hidden_out, caches = r.picker_stream_predict(input_wav, caches)
            if valid_phone_out.shape[1] == 0:
                continue
            feature_outputs, picked_phone_out = r.feature_pick(valid_hidden_out, valid_phone_out)
            if feature_outputs.shape[1] != 0:
                valid_ctc_out, unvalid_txt_outs, caches2 = r.decoder_stream_predict(feature_outputs, caches2)
                valid_txt_outs = torch.cat([valid_txt_outs, valid_ctc_out], 1)
                valid_phone_outs = torch.cat([valid_phone_outs, picked_phone_out], 1)
            txt_output = torch.cat([valid_txt_outs, unvalid_txt_outs], 1)
            if txt_output.shape[1] == 0 or valid_phone_outs.shape[1] == 0:
                continue
            text = self.text_featurizer.iextract(_ctc_text(txt_output, Vt - 1))
            phone = self.phone_featurizer.iextract(_ctc_text(valid_phone_outs, Vp - 1))
            out.append((e / self.speech_featurizer.sample_rate, " ".join(phone), "".join(text)))
            if verbose:
                print("time:", out[-1][0])
                print("streaming phone out:", phone)
                print("streaming texts out:", text)
        offline = self.offline_stt(wav_path)
        if verbose:
            print("offline texts out:", offline)
        return {"streaming": out, "offline": offline}

    stt = offline_stt


class ChunkAMTester(ChunkASR):
    """asr/tester/chunk_tester.py: `runner.predict(features)` -> softmax -> greedy ctc_decode over all picked frames ->
    SER / CER of the text ids with the S / I / D accumulators.  Batches are the 6-tuples of the reference's chunk loader
    (features, input_length, phone_labels, phone_label_length, tar_label, tar_label_length); 5-tuples of EvalList work too."""

    def __init__(self, config, device="cuda:0", load_checkpoint=True):
        self.eval_metrics = {"ser": _Mean(), "cer": _Mean()}
        self.ctc_nums = [0, 0, 0, 0]
        self.steps, self.all_steps = 0, 0
        self.eval_datasets = None
        super().__init__(config, device=device, load_checkpoint=load_checkpoint)

    def set_all_steps(self, all_steps):
        self.all_steps = all_steps

    def set_datasets(self, evaldataset):
        self.eval_datasets = evaldataset

    def _eval_step(self, batch):
        features, tar_label = batch[0], batch[4]
        logits, _ = self.runner.predict(features)
        B, Tp, V = logits.shape
        if Tp == 0:
            hyps = [[] for _ in range(B)]
        else:
            # new_inp_length = ones * ctc_output.shape[1] (chunk_tester.py:40): every picked row, zero padding included
            ids, lens = ctc_greedy_decode(frame_argmax(logits), None, blank=V - 1)
            ids = ids.clamp(min=0).cpu().numpy()
            hyps = [ids[b, :max(int(lens.max().item()), 1)].tolist() for b in range(B)]
        pad = self.text_featurizer.pad
        for hyp, ref in zip(hyps, np.asarray(tar_label)):
            i = [int(t) for t in hyp if int(t) != pad]
            j = [int(t) for t in np.asarray(ref).flatten() if int(t) != pad]
            _, ws, wd, wi = wer(j, i)
            self.ctc_nums[0] += len(j); self.ctc_nums[1] += ws; self.ctc_nums[2] += wi; self.ctc_nums[3] += wd
            self.eval_metrics["ser"].update_state(0 if i == j else 1)
            self.eval_metrics["cer"].reset_states()
            self.eval_metrics["cer"].update_state(sum(self.ctc_nums[1:]) / (self.ctc_nums[0] + 1e-6))

    def results(self):
        r = {k: v.result() for k, v in self.eval_metrics.items()}
        r["s_i_d"] = "{}_{}_{}".format(*self.ctc_nums[1:])
        r["steps"] = self.steps
        return r

    def run(self):
        if self.eval_datasets is None:
            raise RuntimeError("call set_datasets(...) first")
        for batch in self.eval_datasets:
            self._eval_step(batch)
            self.steps += 1
            logging.info("[Eval] [Step %d] %s", self.steps, self.results())
            if self.all_steps and self.steps >= self.all_steps:
                break
        return self.results()
