"""The reference's synthetic "train-to-imitate" data pipeline (recommendation/data.py) on the
MI355X forward path, seeded instead of time-seeded so that a run can be repeated.

  create_data          random (user, country, 6 colours + click, 6 colours + click) records   data.py:81-108
  gen_sample_data      a randomly initialised TEACHER model ranks colours for each record:
                       user country top6-sorted top1-or(-1) top1-probability                     data.py:164-193
  gen_trained_data     consecutive records of the same user (after a stable sort by user) are
                       paired into one 6-field record                                            data.py:239-256
  gen_training_sample  the teacher labels each paired record with its top-1 colour, or -1 where
                       the probability is not above 0.3; the labelled records are shuffled       data.py:196-236
The student then learns to imitate the teacher from train_sample_<i>.csv.  File names and record
formats are the reference's; the arithmetic runs through RecommendationModel on the GPU.
"""
import os
import random

import torch

from .r_model import COLUMNS, RecommendationModel, Sample

threshold = 0.3        # recommendation/data.py:27


def _features(columns):
    return {k: columns[i] for i, k in enumerate(COLUMNS[:6])}


class SampleData(object):
    def __init__(self, user_count, country_count, colour_count, select_count, seed=0):
        self.user_count, self.country_count = user_count, country_count
        self.colour_count, self.select_count = colour_count, select_count
        self.rng = random.Random(seed)

    def random_colours(self):
        colours = self.rng.sample(range(self.colour_count), self.select_count)
        click = self.rng.sample(colours + [-1], 1)
        return colours, click

    def create_data(self, num, output_dir):
        user_dict = {i: self.rng.randint(0, self.country_count - 1) for i in range(self.user_count)}
        os.makedirs(output_dir, exist_ok=True)
        with open(os.path.join(output_dir, 'users.csv'), 'w') as f:
            for i in range(self.user_count):
                f.write('%d %d\n' % (i, user_dict[i]))
        with open(os.path.join(output_dir, 'sample.csv'), 'w') as f:
            for _ in range(num):
                uid = self.rng.randint(0, self.user_count - 1)
                c1, c2 = self.random_colours(), self.random_colours()
                f.write('%d %d %s %d %s %d\n' % (uid, user_dict[uid], ','.join(str(x) for x in sorted(c1[0])), c1[1][0],
                                                 ','.join(str(x) for x in sorted(c2[0])), c2[1][0]))
        return user_dict


def gen_sample_data(teacher, data_dir, index=1, batch_size=4096):
    """sample.csv -> org_sample_<index>.csv through the teacher's top-6 head."""
    with open(os.path.join(data_dir, 'org_sample_%d.csv' % index), 'w') as f:
        for columns in Sample.read_csv(os.path.join(data_dir, 'sample.csv'), batch_size=batch_size):
            with torch.no_grad():
                out = teacher.forward(teacher.features(_features(columns)))
            idx, val = teacher.output(out)
            idx, val = idx.cpu().tolist(), val.cpu().tolist()
            for u, c, ii, vv in zip(columns[0].tolist(), columns[1].tolist(), idx, val):
                f.write('%d %d %s %d %s\n' % (u, c, ','.join(str(x) for x in sorted(ii)),
                                              ii[0] if vv[0] > threshold else -1, repr(float(vv[0]))))


def gen_trained_data(data_dir, index=1):
    """org_sample_<index>.csv -> no_label_sample_<index>.csv: after a stable sort by user, every
    second record of a user closes a pair with the one before it."""
    rows = [l.rstrip('\n').split(' ') for l in open(os.path.join(data_dir, 'org_sample_%d.csv' % index))]
    rows.sort(key=lambda r: int(r[0]))
    key, held = None, None
    with open(os.path.join(data_dir, 'no_label_sample_%d.csv' % index), 'w') as f:
        for r in rows:
            if r[0] == key:
                f.write(' '.join([r[0], r[1], r[2], r[3], held[2], held[3]]) + '\n')
                key, held = None, None
            else:
                key, held = r[0], r
    return os.path.join(data_dir, 'no_label_sample_%d.csv' % index)


def gen_training_sample(teacher, data_dir, index=1, batch_size=4096, seed=0):
    """no_label_sample_<index>.csv -> train_sample_<index>.csv with the teacher's thresholded top-1 label."""
    out = []
    for columns in Sample.read_csv(os.path.join(data_dir, 'no_label_sample_%d.csv' % index), batch_size=batch_size):
        with torch.no_grad():
            logits = teacher.forward(teacher.features(_features(columns)))
        idx, val = teacher.top_one_output(logits)
        idx, val = idx.reshape(-1).cpu().tolist(), val.reshape(-1).cpu().tolist()
        cols = [c.tolist() for c in columns]
        for i in range(len(idx)):
            out.append('%d %d %s %d %s %d %d' % (cols[0][i], cols[1][i], ','.join(str(x) for x in sorted(cols[2][i])),
                                                 cols[3][i], ','.join(str(x) for x in sorted(cols[4][i])), cols[5][i],
                                                 idx[i] if val[i] > threshold else -1))
    random.Random(seed).shuffle(out)
    path = os.path.join(data_dir, 'train_sample_%d.csv' % index)
    with open(path, 'w') as f:
        f.write('\n'.join(out) + '\n')
    return path


def pipeline(data_dir, num=100000, device=None, seed=0, index=1):
    """data.py:282-296 for one teacher: returns (teacher, path of train_sample_<index>.csv)."""
    SampleData(user_count=100, country_count=20, colour_count=128, select_count=6, seed=seed).create_data(num, data_dir)
    teacher = RecommendationModel(colour_count=128, recommend_num=6, user_count=100, country_count=20, device=device,
                                  seed=seed + 1000 * index)
    gen_sample_data(teacher, data_dir, index)
    gen_trained_data(data_dir, index)
    return teacher, gen_training_sample(teacher, data_dir, index, seed=seed)
