"""Batch evaluation harness: the reference's `AMTester` (asr/tester/am_tester.py:13-183, base_tester.py:19-95),
its list-file eval loader conventions (asr/dataloaders/am_dataloader.py:118-227) and its error-rate arithmetic
(utils/xer.py:12-36, 211-220), on the MI355X models of this package.

    tester = AMTester(UserConfig(data_yml, model_yml))
    tester.set_datasets(EvalList(config, tester.speech_featurizer, tester.phone_featurizer, tester.text_featurizer))
    metrics = tester.run()      # {'phone_ser', 'phone_cer', 'txt_ser', 'txt_cer', 'phone_s_i_d', 'trans_s_i_d', ...}

List file lines are `wav_path<TAB>text` as in the reference; the phone labels come from `text_to_vocab(text)`
(the reference uses pypinyin, am_dataloader.py:69-79 -- pass your own callable) or from an optional third
tab-separated column of space-separated phone tokens."""
import logging
import os

import numpy as np

from .asr import ASR
from .models import ctc_greedy_decode


# ---- utils/xer.py ---------------------------------------------------------------------------------------------
def levenshtein(u, v):
    """Edit distance of `v` (hypothesis) against `u` (reference) with the (SUB, DEL, INS) counts of one optimal
    alignment.  Tie-breaking as utils/xer.py:12-36: a substitution/match step is preferred over a deletion, a
    deletion over an insertion -- the S/I/D split (not the distance) depends on it."""
    n, m = len(u), len(v)
    cost = list(range(m + 1))
    ops = [(0, 0, j) for j in range(m + 1)]
    for x in range(1, n + 1):
        new_cost = [x] + [0] * m
        new_ops = [(0, x, 0)] + [None] * m
        ux = u[x - 1]
        for y in range(1, m + 1):
            neq = int(ux != v[y - 1])
            c_sub, c_del, c_ins = cost[y - 1] + neq, cost[y] + 1, new_cost[y - 1] + 1
            best = min(c_sub, c_del, c_ins)
            new_cost[y] = best
            if best == c_sub:
                s, d, i = ops[y - 1]
                new_ops[y] = (s + neq, d, i)
            elif best == c_del:
                s, d, i = ops[y]
                new_ops[y] = (s, d + 1, i)
            else:
                s, d, i = new_ops[y - 1]
                new_ops[y] = (s, d, i + 1)
        cost, ops = new_cost, new_ops
    return cost[m], ops[m]


def wer(r, h):
    """utils/xer.py:211-220: ((S+I+D)/N, S, D, I) for reference `r`, hypothesis `h` (sequences of tokens/ids)."""
    _, (s, d, i) = levenshtein(r, h)
    return (s + i + d) / len(r), s, d, i


class _Mean:
    """tf.keras.metrics.Mean: running mean of the values given to update_state."""

    def __init__(self):
        self.reset_states()

    def reset_states(self):
        self.total, self.count = 0.0, 0

    def update_state(self, v):
        self.total += float(v)
        self.count += 1

    def result(self):
        return self.total / self.count if self.count else 0.0


# ---- asr/dataloaders/am_dataloader.py:118-227 (eval half) -----------------------------------------------------
class EvalList:
    """Iterates a test list in batches the way `AM_DataLoader.eval_data_generator` builds them:
    skip unreadable / shorter than 400 samples / longer than `wav_max_duration`; offline: peak-normalise
    `x / max|x|`, `in_len = len // (reduction * hop)`, zero-pad to the batch maximum; streaming: raw samples, padded
    to the next multiple of the block, `in_len` counted in whole blocks (13 frames each at 0.5 s)."""

    def __init__(self, config, speech_featurizer, phone_featurizer, text_featurizer, test_list=None,
                 text_to_vocab=None, batch_size=None):
        self.speech_config = config["speech_config"]
        self.speech_featurizer, self.phone_featurizer, self.text_featurizer = speech_featurizer, phone_featurizer, text_featurizer
        self.batch = batch_size or config["running_config"]["batch_size"]
        self.streaming = self.speech_config["streaming"]
        self.chunk = int(self.speech_config["sample_rate"] * self.speech_config["streaming_bucket"])
        self.text_to_vocab = text_to_vocab
        path = test_list or self.speech_config["eval_list"]
        with open(path, encoding="utf-8") as f:
            self.test_list = [l.strip() for l in f.readlines() if l.strip() != ""]
        self.test_offset = 0

    def eval_per_epoch_steps(self):
        return len(self.test_list) // self.batch

    @staticmethod
    def only_chinese(word):
        return "".join(ch for ch in word if "一" <= ch <= "鿿")

    @staticmethod
    def check_valid(txt, vocab_list):
        if len(txt) == 0:
            return False
        for n in txt:
            if n not in vocab_list:
                return n
        return True

    def _phones_of(self, fields, txt):
        if len(fields) > 2:
            return fields[2].split()
        if self.text_to_vocab is None:
            raise ValueError("list line has no phone column and no text_to_vocab callable was given "
                             "(the reference derives phones with pypinyin, which this image does not ship)")
        return self.text_to_vocab(txt)

    def eval_data_generator(self):
        sc = self.speech_config
        reduce = sc["reduction_factor"] * (self.speech_featurizer.sample_rate / 1000) * sc["stride_ms"]
        feats, input_length, phones, phones_length, txts = [], [], [], [], []
        max_input = 0
        for _ in range(self.batch * 10):
            line = self.test_list[self.test_offset]
            self.test_offset += 1
            if self.test_offset > len(self.test_list) - 1:
                self.test_offset = 0
            fields = line.strip().split("\t")
            wp, txt = fields[0], fields[1]
            try:
                data = self.speech_featurizer.load_wav(wp)
            except Exception:
                logging.info("%s load data failed,skip", wp)
                continue
            if len(data) < 400:
                continue
            if len(data) > self.speech_featurizer.sample_rate * sc["wav_max_duration"]:
                logging.info("%s duration out of wav_max_duration(%s),skip", wp, sc["wav_max_duration"])
                continue
            if sc.get("only_chinese"):
                txt = self.only_chinese(txt)
            if not self.streaming:
                feat = data / np.abs(data).max()
                in_len = len(feat) // reduce
            else:
                feat = data
                in_len = len(feat) // self.chunk + (1 if len(feat) % self.chunk != 0 else 0)
                chunk_times = self.chunk // reduce + (1 if self.chunk % reduce != 0 else 0)
                in_len *= chunk_times
            py = self._phones_of(fields, txt)
            if self.check_valid(py, self.phone_featurizer.vocab_array) is not True:
                logging.info(" %s phones not all in tokens,continue", txt)
                continue
            if self.check_valid(txt, self.text_featurizer.vocab_array) is not True:
                logging.info(" %s text not all in tokens,continue", txt)
                continue
            phone_feature = self.phone_featurizer.extract(py)
            text_feature = self.text_featurizer.extract(list(txt)) + [self.text_featurizer.endid()]
            if in_len < len(phone_feature):
                logging.info("%s feature length < phone length,continue", wp)
                continue
            max_input = max(max_input, len(feat))
            feats.append(feat)
            input_length.append(in_len)
            phones.append(phone_feature)
            txts.append(text_feature)
            phones_length.append(len(phone_feature))
            if len(feats) == self.batch:
                break
        if not feats:
            raise RuntimeError("no usable utterance in the evaluation list")
        if self.streaming:
            # am_dataloader.py:198-209 rounds max_input up to the next block boundary TWICE (before each of its two
            # pad_signal calls), so every streaming batch carries one extra all-zero block; the CTCDecoder and the
            # Translator attend over all frames without a mask, so that block is part of the reference's numbers
            max_input = max_input // self.chunk * self.chunk + self.chunk
            max_input = max_input // self.chunk * self.chunk + self.chunk
            chunk_times = self.chunk // reduce + (1 if self.chunk % reduce != 0 else 0)
            input_length = np.clip(input_length, 0, (max_input // self.chunk) * chunk_times)
        x = self.speech_featurizer.pad_signal(feats, max_input)[..., None]

        def pad(seqs, value):
            out = np.full((len(seqs), max(len(s) for s in seqs)), value, np.int32)
            for i, s in enumerate(seqs):
                out[i, :len(s)] = s
            return out
        return (x.astype(np.float32), np.array(input_length, "int32"), pad(phones, self.phone_featurizer.pad),
                np.array(phones_length, "int32"), pad(txts, self.text_featurizer.pad))

    def __iter__(self):
        while True:
            yield self.eval_data_generator()


# ---- asr/tester/am_tester.py ----------------------------------------------------------------------------------
class AMTester(ASR):
    """`AMTester(config)`: same models / checkpoints as `ASR`, evaluated over batches
    `(features [B,L,1], input_length [B], phone_labels [B,P], phone_label_length [B], text_labels [B,Q])`."""

    def __init__(self, config, device="cuda:0", load_checkpoint=True):
        self.config = config
        self.eval_metrics = {k: _Mean() for k in ("phone_ser", "phone_cer", "txt_ser", "txt_cer")}
        self.ctc_nums = [0, 0, 0, 0]          # n, s, i, d
        self.translator_nums = [0, 0, 0, 0]
        self.steps, self.all_steps = 0, 0
        self.eval_datasets = None
        super().__init__(config, device=device, load_checkpoint=load_checkpoint)
        self.output_file_path = os.path.join(self.running_config["outdir"], "test.tsv")

    def set_all_steps(self, all_steps):
        self.all_steps = all_steps

    def set_datasets(self, evaldataset):
        self.eval_datasets = evaldataset

    def finished(self):
        return self.steps >= self.all_steps

    @staticmethod
    def _strip(seq, values):
        return [int(t) for t in seq if int(t) not in values]

    def _eval_step(self, batch):
        """am_tester.py:34-89."""
        features, input_length, phone_labels, _, tar_label = batch
        enc_output = self.encoder(features, training=False)
        _, frame_ids = self.ctc_model(enc_output, training=False, return_argmax=True, return_logits=False)
        # tf.keras.backend.ctc_decode treats the LAST class as the blank whatever `blank_at_zero` says (am_tester.py:38-40)
        ids, lens = ctc_greedy_decode(frame_ids, input_length, blank=self.phone_featurizer.num_classes - 1)
        ctc_decode = ids[:, :max(int(lens.max().item()), 1)].clamp_(min=0).contiguous()
        _, translator_out = self.translator([ctc_decode, enc_output], training=False, return_argmax=True)
        ctc_decode, translator_out = ctc_decode.cpu().numpy(), translator_out.cpu().numpy()
        pad = self.phone_featurizer.pad
        for hyp, ref in zip(ctc_decode, np.asarray(phone_labels)):
            i, j = self._strip(hyp, (pad,)), self._strip(ref, (pad,))
            _, ws, wd, wi = wer(j, i)
            self.ctc_nums[0] += len(j); self.ctc_nums[1] += ws; self.ctc_nums[2] += wi; self.ctc_nums[3] += wd
            self.eval_metrics["phone_ser"].update_state(0 if i == j else 1)
            self.eval_metrics["phone_cer"].reset_states()
            self.eval_metrics["phone_cer"].update_state(sum(self.ctc_nums[1:]) / (self.ctc_nums[0] + 1e-6))
        tpad, tend = self.text_featurizer.pad, self.text_featurizer.endid()
        for hyp, ref in zip(translator_out, np.asarray(tar_label)):
            hyp = [int(t) for t in hyp]
            if 1 in hyp:
                hyp = hyp[:hyp.index(1)]
            i, j = self._strip(hyp, (tpad, tend)), self._strip(ref, (tpad, tend))
            _, ws, wd, wi = wer(j, i)
            self.translator_nums[0] += len(j); self.translator_nums[1] += ws
            self.translator_nums[2] += wi; self.translator_nums[3] += wd
            self.eval_metrics["txt_ser"].update_state(0 if i == j else 1)
            self.eval_metrics["txt_cer"].reset_states()
            self.eval_metrics["txt_cer"].update_state(sum(self.translator_nums[1:]) / (self.translator_nums[0] + 1e-6))

    def results(self):
        r = {k: v.result() for k, v in self.eval_metrics.items()}
        r["phone_s_i_d"] = "{}_{}_{}".format(*self.ctc_nums[1:])
        r["trans_s_i_d"] = "{}_{}_{}".format(*self.translator_nums[1:])
        r["steps"] = self.steps
        return r

    def _eval_batches(self):
        for batch in self.eval_datasets:
            self._eval_step(batch)
            self.steps += 1
            logging.info("[Eval] [Step %d] %s", self.steps, self.results())
            if self.finished():
                break

    def run(self):
        if self.eval_datasets is None:
            raise RuntimeError("call set_datasets(...) first")
        if not self.all_steps:
            self.set_all_steps(max(self.eval_datasets.eval_per_epoch_steps(), 1)
                               if hasattr(self.eval_datasets, "eval_per_epoch_steps") else 1)
        self._eval_batches()
        return self.results()
