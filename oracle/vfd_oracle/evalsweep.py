"""CPU restatement of the reference's in-loop evaluation sweeps (test infrastructure: the checker of the HIP-side test()).

  mygan_test   reference models/mygannet.py:369-475 (nets as they are — the reference never calls .eval() there —, threshold,
               5x5 opening, both discriminator passes, 12 loss means, ROC / PR / F1)
  anogan_test  reference models/anogan.py:145-227 (nets in eval mode, predict_forg :24-37)
  morph_open5  cv2.morphologyEx(MORPH_OPEN, ones((5,5))) of lib/utils.py:139-147 per frame, restated with scipy.ndimage
               min / max filters with an infinite constant border (cv2's default morphology border: outside pixels do not
               take part).  cv2 is absent from this image, so THIS restatement is unpinned against cv2 itself; it follows
               cv2's documented semantics.
  scores       the sklearn calls of lib/evaluate.py:14-91
"""
import numpy as np
import scipy.ndimage as ndi
import torch
import torch.nn as nn
from sklearn.metrics import auc, f1_score, precision_recall_curve, roc_curve

from .losses import l2_loss, weighted_bce


def morph_open5(video):
    """(…, H, W) float array -> opened array."""
    v = np.asarray(video, dtype=np.float32)
    flat = v.reshape((-1,) + v.shape[-2:])
    out = np.empty_like(flat)
    for i, img in enumerate(flat):
        er = ndi.minimum_filter(img, size=5, mode="constant", cval=np.inf)
        out[i] = ndi.maximum_filter(er, size=5, mode="constant", cval=-np.inf)
    return out.reshape(v.shape)


def scores(gts, predicts):
    gts = np.asarray(gts, dtype=np.int32).flatten()
    predicts = np.asarray(predicts, dtype=np.float32).flatten().copy()
    fpr, tpr, _ = roc_curve(gts, predicts)
    precision, recall, _ = precision_recall_curve(gts, predicts)
    roc, pr = auc(fpr, tpr), auc(recall, precision)
    predicts[predicts >= 0.2] = 1
    predicts[predicts < 0.2] = 0
    return {"roc": roc, "pr": pr, "f1": f1_score(gts, predicts)}


def mygan_test(netg, netd, batches, flows, w_adv=1, w_con=10):
    """batches: list of (input, real, gt, lb); flows: list of (gt_flow, pre_flow)."""
    l_bce = nn.BCELoss()
    acc = {k: [] for k in ("err_g_adv_s", "err_g_adv_t", "err_g_con", "err_d_real_s", "err_d_real_t", "err_d_fake_s", "err_d_fake_t")}
    gts, predicts = [], []
    with torch.no_grad():
        for (inp, real, gt, lb), (gf, pf) in zip(batches, flows):
            b = inp.shape[0]
            ones, zeros = torch.ones(b), torch.zeros(b)
            predict = netg(inp)                                                   # :394
            t_pre = (predict > 0.5).float()                                       # :395 threshold
            m_pre = torch.from_numpy(morph_open5(t_pre.numpy()))                  # :396
            gts.append(gt.permute(0, 2, 3, 4, 1).numpy())
            predicts.append(m_pre.permute(0, 2, 3, 4, 1).numpy())
            gt3, pre3 = torch.cat([gt] * 3, dim=1), torch.cat([predict] * 3, dim=1)     # :401-402
            s_pr, s_fr, t_pr, t_fr = netd(gt3, gf)                                # :406-409
            s_pf, s_ff, t_pf, t_ff = netd(pre3, pf)
            acc["err_g_adv_s"].append(l2_loss(s_fr, s_ff).item())                 # :411-415
            acc["err_g_adv_t"].append(l2_loss(t_fr, t_ff).item())
            acc["err_g_con"].append(weighted_bce(predict, gt).item())
            acc["err_d_real_s"].append(l_bce(s_pr, ones).item())                  # :418-424
            acc["err_d_real_t"].append(l_bce(t_pr, ones).item())
            acc["err_d_fake_s"].append(l_bce(s_pf, zeros).item())
            acc["err_d_fake_t"].append(l_bce(t_pf, zeros).item())
    e = {k: np.asarray(v, dtype=np.float64) for k, v in acc.items()}
    err_d_real, err_d_fake = (e["err_d_real_s"] + e["err_d_real_t"]) * 0.5, (e["err_d_fake_s"] + e["err_d_fake_t"]) * 0.5
    out = {'d/err_d_real_s/test': e["err_d_real_s"].mean(), 'd/err_d_real_t/test': e["err_d_real_t"].mean(),
           'd/err_d_fake_s/test': e["err_d_fake_s"].mean(), 'd/err_d_fake_t/test': e["err_d_fake_t"].mean(),
           'd/err_d_real/test': err_d_real.mean(), 'd/err_d_fake/test': err_d_fake.mean(),
           'd/err_d/test': ((err_d_real + err_d_fake) * 0.5).mean(),
           'g/err_g_adv_s/test': e["err_g_adv_s"].mean(), 'g/err_g_adv_t/test': e["err_g_adv_t"].mean(),
           'g/err_g_adv/test': (e["err_g_adv_s"] + e["err_g_adv_t"]).mean(), 'g/err_g_con/test': e["err_g_con"].mean(),
           'g/err_g/test': (e["err_g_adv_t"] * w_adv + e["err_g_con"] * w_con).mean()}          # :416: temporal term only
    return {k: float(v) for k, v in out.items()}, scores(np.stack(gts), np.stack(predicts))


def _normalize(t):                                                                # lib/utils.py:81-89
    mn, mx = float(t.min()), float(t.max())
    return (t.clamp(min=mn, max=mx) - mn) / (mx - mn + 1e-5)


def predict_forg(gout, inp):                                                      # models/anogan.py:24-37
    diff = torch.abs(gout - inp)
    frames = torch.stack([_normalize(v) for v in diff.permute(2, 0, 1, 3, 4)])    # per time step over (B,C,H,W)
    frames = frames.permute(1, 2, 0, 3, 4)                                        # (B,C,T,H,W)
    return 0.299 * frames[:, 0:1] + 0.587 * frames[:, 1:2] + 0.114 * frames[:, 2:3]   # cv2.COLOR_RGB2GRAY


def anogan_test(netg, netd, batches, zs):
    loss = nn.BCELoss()
    netg.eval(); netd.eval()                                                      # :146-147
    gen_l, dr, df, gts, predicts = [], [], [], [], []
    with torch.no_grad():
        for (inp, real, gt, lb), z in zip(batches, zs):
            b = real.shape[0]
            ones, zeros = torch.ones(b), torch.zeros(b)
            dr.append(loss(netd(real)[0].view(-1), ones).item())                  # :166-167
            gen_fake = netg(z)                                                    # :170
            dis_fake = netd(gen_fake)[0].view(-1)
            df.append(loss(dis_fake, zeros).item())                               # :172
            gen_l.append(loss(dis_fake, ones).item())                             # :176-177
            predict = predict_forg(gen_fake, real)                                # :179
            gts.append(gt.permute(0, 2, 3, 4, 1).numpy())
            predicts.append(predict.permute(0, 2, 3, 4, 1).numpy())
    return ({"gen_loss": float(np.mean(gen_l)), "dis_loss_real": float(np.mean(dr)), "dis_loss_fake": float(np.mean(df))},
            scores(np.stack(gts), np.stack(predicts)))


def ganomaly_test(netg, batches, fold):
    """models/ganomaly.py:332-406: per test sample the anomaly score mean((latent_i - latent_o)^2) over the nz latent
    channels (:370-372), min-max scaled over the whole test set (:396), ROC AUC against the labels (:398).  The reference
    does NOT switch netg to eval mode here (no .eval() in the file): BatchNorm normalises with the statistics of each test
    batch and keeps updating its running statistics, under torch.no_grad().  `fold`: clip block -> frames (the 4-tuple clip
    contract of lib/train_gan.py:69 in front of the 2-D nets); every frame carries its clip's label."""
    an, lab = [], []
    with torch.no_grad():
        for (inp, real, gt, lb) in batches:
            x = fold(inp)
            fake, latent_i, latent_o = netg(x)
            err = torch.mean(torch.pow(latent_i - latent_o, 2), dim=1)                  # :372
            an.append(err.reshape(err.size(0)))
            lab.append(lb.reshape(-1).repeat_interleave(x.shape[0] // lb.numel()))
    an = torch.cat(an)
    lab = torch.cat(lab).long()
    an = (an - an.min()) / (an.max() - an.min())                                        # :396
    fpr, tpr, _ = roc_curve(lab.numpy(), an.numpy())
    return {"AUC": float(auc(fpr, tpr)), "an_scores": an.numpy(), "gt_labels": lab.numpy()}
