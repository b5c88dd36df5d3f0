#!/usr/bin/env python
"""SqueezeDet image demo on MI355X: the py3 / HIP counterpart of the reference's src/demo.py image_demo
(:160-225) -- same call shape (`sess.run([det_boxes, det_probs, det_class], {image_input: [im]})`,
`model.filter_prediction`, PLOT_PROB_THRESH, class colours) with the image preparation (float cast, bilinear
resize to the network input, BGR mean subtraction, demo.py:186-190) done by sqdet_preprocess_bgr on the GPU.
cv2 is replaced by PIL for file I/O and drawing.

    python demo.py --input_path 'data/*.png' --out_dir out/ [--weights weights.npz] [--demo_net squeezeDet]

--weights: a {variable name: array} file written by squeezedet_amd.weights.save_params (or converted from a
reference checkpoint with squeezedet_amd.weights.from_reference_names); without it seeded synthetic weights
are used (there is no network access to fetch the reference's checkpoint), so the boxes are meaningless but
the whole path runs.
"""
import argparse
import glob
import os

import numpy as np
import torch


def draw_boxes(img_rgb, boxes, labels, cdict):
    """_draw_box of src/demo.py / src/train.py:51-72 with PIL: cx,cy,w,h boxes, label at the top-left corner."""
    from PIL import ImageDraw
    d = ImageDraw.Draw(img_rgb)
    for (cx, cy, w, h), lab in zip(boxes, labels):
        x1, y1, x2, y2 = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2
        bgr = cdict.get(lab.split(":")[0], (0, 255, 0))
        d.rectangle([x1, y1, x2, y2], outline=(bgr[2], bgr[1], bgr[0]), width=2)
        d.text((x1 + 2, y1 + 2), lab, fill=(bgr[2], bgr[1], bgr[0]))
    return img_rgb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--input_path", default="./data/sample.png", help="glob of input images")
    ap.add_argument("--out_dir", default="./data/out/")
    ap.add_argument("--demo_net", default="squeezeDet", choices=["squeezeDet", "squeezeDet+", "resnet50"])
    ap.add_argument("--weights", default="")
    ap.add_argument("--gpu", default="0")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    a = ap.parse_args()
    from PIL import Image
    import squeezedet_amd as S
    from squeezedet_amd import nets, ops, synthetic, weights
    from squeezedet_amd.nn_skeleton import Session
    mc, cls = {"squeezeDet": (S.kitti_squeezeDet_config, nets.SqueezeDet), "squeezeDet+": (S.kitti_squeezeDetPlus_config, nets.SqueezeDetPlus),
               "resnet50": (S.kitti_res50_config, nets.ResNet50ConvDet)}[a.demo_net]
    mc = mc()
    mc.BATCH_SIZE = 1
    mc.LOAD_PRETRAINED_MODEL = False          # parameters are restored below (demo.py:171-172)
    dtype = torch.float16 if a.dtype == "fp16" else torch.float32
    model = cls(mc, a.gpu, dtype=dtype)
    model.load_params(weights.load_params(a.weights) if a.weights else synthetic.synthetic_params(model, seed=0))
    os.makedirs(a.out_dir, exist_ok=True)
    cls2clr = {"car": (255, 191, 0), "cyclist": (0, 191, 255), "pedestrian": (255, 0, 191)}
    with Session() as sess:
        for f in glob.iglob(a.input_path):
            rgb = np.asarray(Image.open(f).convert("RGB"))
            bgr = torch.from_numpy(np.ascontiguousarray(rgb[:, :, ::-1])).to(model.device)       # what cv2.imread returns
            input_image = ops.preprocess_bgr(bgr[None], mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, mc.BGR_MEANS, dtype)
            det_boxes, det_probs, det_class = sess.run([model.det_boxes, model.det_probs, model.det_class],
                                                       feed_dict={model.image_input: input_image})
            final_boxes, final_probs, final_class = model.filter_prediction(det_boxes[0], det_probs[0], det_class[0])
            keep = [i for i in range(len(final_probs)) if final_probs[i] > mc.PLOT_PROB_THRESH]
            labels = [mc.CLASS_NAMES[final_class[i]] + ": (%.2f)" % final_probs[i] for i in keep]
            # boxes are in network-input coordinates: draw on the resized image like the reference does
            im = Image.fromarray(rgb).resize((mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT), Image.BILINEAR)
            draw_boxes(im, [final_boxes[i] for i in keep], labels, cls2clr)
            out = os.path.join(a.out_dir, "out_" + os.path.split(f)[1])
            im.save(out)
            print("Image detection output saved to {} ({} boxes above {:.2f})".format(out, len(keep), mc.PLOT_PROB_THRESH))


if __name__ == "__main__":
    main()
