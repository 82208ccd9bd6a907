"""CPU oracle for metrics.SELDMetrics (metrics.py:7-170), numpy fp64, statement by statement.
TEST INFRASTRUCTURE ONLY (see oracle/seldnet_oracle.py header).  The reference has no test for this
class, so its numerics are pinned only by this restatement and the hand-worked case in
tests/test_oracle_pins.py."""
import numpy as np


def safe_div(x, y, eps=1e-8):
    return x / np.maximum(y, eps)


def l2_normalize(x, axis=-1, eps=1e-12):
    return x / np.sqrt(np.maximum((x ** 2).sum(axis, keepdims=True), eps))


def distance_between_cartesian_coordinates(xyz0, xyz1):
    xyz0, xyz1 = l2_normalize(xyz0), l2_normalize(xyz1)
    zeros = (xyz0.sum(-1) == 0).astype(float) * (xyz1.sum(-1) == 0).astype(float)
    d = np.clip((xyz0 * xyz1).sum(-1), -1, 1)
    return np.arccos(d) / np.pi * 180 * (1 - zeros)


class SELDMetrics:
    def __init__(self, doa_threshold=20, block_size=10, n_classes=14):
        self.doa_threshold, self.block_size, self.n_classes = doa_threshold, block_size, n_classes
        self.reset_states()

    def reset_states(self):
        for k in ("TP", "FP", "TN", "FN", "S", "D", "I", "Nref", "Nsys", "total_DE", "DE_TP"):
            setattr(self, k, 0.0)
        for k in ("class_tp", "class_fp", "class_tn", "class_fn"):
            setattr(self, k, np.zeros(self.n_classes))

    def result(self):
        ER = safe_div(self.S + self.D + self.I, self.Nref)
        prec, recall = safe_div(self.TP, self.TP + self.FP), safe_div(self.TP, self.TP + self.FN)
        F = safe_div(2 * prec * recall, prec + recall)
        DE = safe_div(self.total_DE, self.DE_TP) if self.DE_TP > 0 else 180.0
        DE_prec, DE_recall = safe_div(self.DE_TP, self.Nsys), safe_div(self.DE_TP, self.Nref)
        return ER, F, DE, safe_div(2 * DE_prec * DE_recall, DE_prec + DE_recall)

    def state_vector(self):
        return np.concatenate([[self.TP, self.FP, self.TN, self.FN, self.S, self.D, self.I, self.Nref, self.Nsys,
                                self.total_DE, self.DE_TP], self.class_tp, self.class_fp, self.class_tn, self.class_fn])

    def update_states(self, y_true, y_pred):
        sed_t, doa_t = (np.asarray(a, np.float64) for a in y_true)
        sed_p, doa_p = (np.asarray(a, np.float64) for a in y_pred)
        n = sed_t.shape[-2]
        for i in range((n + self.block_size - 1) // self.block_size):
            sl = slice(i * self.block_size, (i + 1) * self.block_size)
            self.update_block_states((sed_t[..., sl, :], doa_t[..., sl, :]), (sed_p[..., sl, :], doa_p[..., sl, :]))

    def update_block_states(self, tb, pb):
        sed_true, doa_true = tb
        sed_pred, doa_pred = pb
        sed_pred = (sed_pred > 0.5).astype(np.float64)
        if sed_true.ndim == 2:
            sed_true, sed_pred, doa_true, doa_pred = sed_true[None], sed_pred[None], doa_true[None], doa_pred[None]
        doa_true = np.swapaxes(doa_true.reshape(*doa_true.shape[:-1], 3, -1), -1, -2)
        doa_pred = np.swapaxes(doa_pred.reshape(*doa_pred.shape[:-1], 3, -1), -1, -2)
        true_classes = sed_true.max(-2, keepdims=True)
        pred_classes = sed_pred.max(-2, keepdims=True)
        self.Nref += true_classes.sum()
        self.Nsys += pred_classes.sum()
        self.TN += ((1 - true_classes) * (1 - pred_classes)).sum()
        false_negative = true_classes * (1 - pred_classes)
        false_positive = (1 - true_classes) * pred_classes
        true_negative = (1 - true_classes) * (1 - pred_classes)
        true_positives = true_classes * pred_classes
        self.class_fn += false_negative.sum((-3, -2))
        self.class_fp += false_positive.sum((-3, -2))
        self.class_tn += true_negative.sum((-3, -2))
        self.class_tp += true_positives.sum((-3, -2))
        self.FN += false_negative.sum()
        self.FP += false_positive.sum()
        loc_FN = false_negative.sum((-2, -1))
        loc_FP = false_positive.sum((-2, -1))
        frames_matched = (sed_true * true_positives) * (sed_pred * true_positives)
        total_matched_frames = frames_matched.sum(-2, keepdims=True)
        matched_frames_exist = (total_matched_frames > 0).astype(np.float64)
        self.DE_TP += matched_frames_exist.sum()
        false_negative = true_positives * (1 - matched_frames_exist)
        self.FN += false_negative.sum()
        loc_FN = loc_FN + false_negative.sum((-2, -1))
        ang = distance_between_cartesian_coordinates(doa_true * frames_matched[..., None], doa_pred * frames_matched[..., None])
        average_distances = safe_div(ang.sum(-2, keepdims=True), total_matched_frames)
        self.total_DE += average_distances.sum()
        close_angles = (average_distances <= self.doa_threshold).astype(np.float64)
        self.TP += (close_angles * matched_frames_exist).sum()
        false_negative = (1 - close_angles) * matched_frames_exist
        self.FN += false_negative.sum()
        loc_FN = loc_FN + false_negative.sum((-2, -1))
        self.S += np.minimum(loc_FP, loc_FN).sum()
        self.D += np.maximum(0, loc_FN - loc_FP).sum()
        self.I += np.maximum(0, loc_FP - loc_FN).sum()


def calculate_seld_score(metric_values):
    er, f, de, rec = metric_values
    return (er + 1 - f + de / 180 + 1 - rec) / 4
