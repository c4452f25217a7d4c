import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_session_gpu as T
for sess in (True, False):
    a, fa, m = T._run_train_mode_steps(4, session=sess)
    st = a.state
    print("session" if sess else "generic", [round(x, 5) for x in st["init_losses"]], [round(x, 5) for x in fa], st["num_cg_iters"], st["best_cg_iters"], st["learning_rates"], [round(d, 4) for d in st["dampings"]], int(m.bn1.num_batches_tracked))
