"""Model-folder naming rule of the reference (``directory.py:31-40``) so ``load()`` finds the same dirs."""
import os


def find_model_folder(FLAGS, postfix):
    name = "result_{}_deep_ld{}_c{}_l{}_cheb{}_diver{}_{}_{}".format(
        FLAGS.training_set, FLAGS.feature_size, FLAGS.hidden1, FLAGS.num_layer, FLAGS.max_degree,
        FLAGS.diver_num, FLAGS.predict, postfix)
    path = os.path.join("./model", name)
    snap = getattr(FLAGS, "snapshot", "")
    if snap:
        path = os.path.join(path, snap)
    return path
