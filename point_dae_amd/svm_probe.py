"""Linear-SVM probe of the pretrained encoder: the validation loop of the pretraining runner.

Reference: tools/runner_pretrain.py:25-48 (Acc_Metric, evaluate_svm), :290-349 (validate): every
labelled cloud of the `extra_train` and `val` loaders is FPS-resampled to `npoints`, the model's
`return_feat` path gives one feature per cloud (max + mean pooled tokens for the Transformer variant,
models/PointCAE_transformer.py:1024-1026; the global max-pooled feature for DGCNN / PointNet++), the
features of all ranks are gathered, a sklearn LinearSVC is fitted on the train features and scored on
the test features.  FPS and the encoders run on the gfx950 kernels; the SVM fit stays on the host as
in the reference.
"""
import numpy as np
import torch

from . import dist_utils, misc


class Acc_Metric:
    def __init__(self, acc=0.):
        self.acc = acc['acc'] if isinstance(acc, dict) else acc

    def better_than(self, other):
        return self.acc > other.acc

    def state_dict(self):
        return {'acc': self.acc}


def evaluate_svm(train_features, train_labels, test_features, test_labels):
    from sklearn.svm import LinearSVC
    clf = LinearSVC()
    clf.fit(train_features, train_labels)
    pred = clf.predict(test_features)
    return float(np.sum(test_labels == pred)) / pred.shape[0]


@torch.no_grad()
def extract_features(base_model, loader, npoints):
    """-> (features (n, C), labels (n,)) on the device; loader items: (taxonomy, model_id, (points, label))."""
    feats, labels = [], []
    for _, _, data in loader:
        points, label = data[0].cuda(), data[1].cuda()
        # ALWAYS resampled (runner_pretrain.py:305,318): with npoints == N the FPS still re-ORDERS the cloud (sample
        # order, starting from point 0), which moves the model's own FPS centres -- skipping it changes the features
        _, points = misc.fps(points, npoints)
        assert points.shape[1] == npoints
        feats.append(base_model(points, points, vis=False, return_feat=True).detach())
        labels.append(label.view(-1).detach())
    return torch.cat(feats, 0), torch.cat(labels, 0)


def _gather(t, world):
    if world == 1:
        return t
    out = [torch.empty_like(t) for _ in range(world)]
    torch.distributed.all_gather(out, t.contiguous())
    return torch.cat(out, 0)


def validate(base_model, extra_train_loader, test_loader, epoch, config, log=print):
    """runner_pretrain.py:290-349 -> Acc_Metric."""
    rank, world = dist_utils.get_dist_info()
    was_training = base_model.training
    base_model.eval()
    npoints = config.dataset.extra_train.others.npoints
    tr_f, tr_l = extract_features(base_model, extra_train_loader, npoints)
    te_f, te_l = extract_features(base_model, test_loader, npoints)
    tr_f, tr_l, te_f, te_l = (_gather(t, world) for t in (tr_f, tr_l, te_f, te_l))
    acc = evaluate_svm(tr_f.cpu().numpy(), tr_l.cpu().numpy(), te_f.cpu().numpy(), te_l.cpu().numpy())
    if rank == 0:
        log('[Validation] EPOCH: %d  acc = %.4f' % (epoch, acc))
    base_model.train(was_training)
    return Acc_Metric(acc)
