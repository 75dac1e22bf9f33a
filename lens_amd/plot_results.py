"""PR / Recall@N comparison artefacts of `--PR_curve --sad` (outputs of lens/tools/plot_results.py:6-59):
Precision_Recall_plot.pdf, PR_curve_LENS.json, PR_curve_SAD.json, SAD_vs_LENS_RecallatN.pdf/.csv."""
import csv
import json
import os


def plot_PR(LENS_PR, SAD_PR, output_path):
    import matplotlib.pyplot as plt
    plt.figure(figsize=(8, 4))
    plt.plot(SAD_PR["Recall"], SAD_PR["Precision"], label="SAD", color="#D753CC")
    plt.plot(LENS_PR["Recall"], LENS_PR["Precision"], label="LENS", color="#575AB1")
    plt.title("Precision-Recall Comparison")
    plt.xlabel("Recall")
    plt.ylabel("Precision")
    plt.legend()
    plt.xlim(0.0, 1.0)
    plt.ylim(0.0, 1.05)
    plt.savefig(os.path.join(output_path, "Precision_Recall_plot.pdf"), dpi=300)
    plt.close()
    for name, data in (("PR_curve_LENS.json", LENS_PR), ("PR_curve_SAD.json", SAD_PR)):
        with open(os.path.join(output_path, name), "w") as fh:
            json.dump(data, fh)


def plot_recall(lens_recall, sad_recall, N, output_path):
    import matplotlib.pyplot as plt
    plt.figure(figsize=(8, 4))
    plt.plot(N, sad_recall, label="SAD", color="#D753CC", marker="o")
    plt.plot(N, lens_recall, label="LENS", color="#575AB1", marker="o")
    plt.title("Comparison of SAD and LENS")
    plt.xlabel("N")
    plt.ylabel("Values")
    plt.ylim(0, 1)
    plt.legend()
    plt.savefig(os.path.join(output_path, "SAD_vs_LENS_RecallatN.pdf"), dpi=300)
    plt.close()
    with open(os.path.join(output_path, "SAD_vs_LENS_RecallatN.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["N", "SAD", "LENS"])
        for n, s, l in zip(N, sad_recall, lens_recall):
            w.writerow([n, s, l])
