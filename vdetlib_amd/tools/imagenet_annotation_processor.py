#!/usr/bin/env python3
"""ILSVRC2015-VID XML annotations of one video -> an .annot protocol dict
(reference tools/imagenet_annotation_processor.py:53-118).

Same output schema ({'video', 'annotations': [{'id', 'track': [{'frame', 'bbox', 'name', 'class',
'class_index', 'generated', 'occluded', 'frame_size'}]}]}), frame = int(<filename>) + 1 (:71),
frame_size = [height, width] (:108).  Parsed with xml.etree (the reference needs xmltodict, which
is not a dependency here).  XML files are read in sorted order (the reference uses glob order)."""
import argparse
import glob
import json
import os
import sys
import xml.etree.ElementTree as ET

from ..vdet.dataset import imagenet_vdet_classes

# WordNet id -> (VID class index, class name); the 30 ILSVRC2015-VID synsets in class-list order
_WNIDS = ['n02691156', 'n02419796', 'n02131653', 'n02834778', 'n01503061', 'n02924116', 'n02958343', 'n02402425',
          'n02084071', 'n02121808', 'n02503517', 'n02118333', 'n02510455', 'n02342885', 'n02374451', 'n02129165',
          'n01674464', 'n02484322', 'n03790512', 'n02324045', 'n02509815', 'n02411705', 'n01726692', 'n02355227',
          'n02129604', 'n04468005', 'n01662784', 'n04530566', 'n02062744', 'n02391049']
name_map = dict((wnid, (i + 1, imagenet_vdet_classes[i + 1])) for i, wnid in enumerate(_WNIDS))


def track_by_id(annotations, track_id):
    tracks = [track for track in annotations if track['id'] == track_id]
    assert len(tracks) <= 1
    return tracks[0] if tracks else []


def annot_proto_from_dir(annot_dir):
    annot = {'video': os.path.basename(os.path.normpath(annot_dir))}
    annotations = []
    for xml_file in sorted(glob.glob(os.path.join(annot_dir, '*.xml'))):
        root = ET.parse(xml_file).getroot()
        frame = int(root.findtext('filename')) + 1
        size = root.find('size')
        frame_width, frame_height = int(size.findtext('width')), int(size.findtext('height'))
        objects = root.findall('object')
        if not objects:
            print("xml {} has no objects.".format(xml_file))
            continue
        for box in objects:
            track_id = str(box.findtext('trackid'))
            track = track_by_id(annotations, track_id)
            if not track:
                track = {"id": track_id, "track": []}
                annotations.append(track)
            bb = box.find('bndbox')
            bbox = [int(bb.findtext(k)) for k in ('xmin', 'ymin', 'xmax', 'ymax')]
            name = str(box.findtext('name'))
            cls_idx, cls_name = name_map[name]
            track['track'].append({"frame": frame, "bbox": bbox, "name": name, "class": cls_name,
                                   "class_index": cls_idx, "generated": int(box.findtext('generated')),
                                   "occluded": int(box.findtext('occluded')),
                                   "frame_size": [frame_height, frame_width]})
    annot['annotations'] = annotations
    return annot


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('annot_dir')
    parser.add_argument('save_file')
    args = parser.parse_args(argv)
    if os.path.isfile(args.save_file):
        print("{} already exists.".format(args.save_file))
        return 0
    annot = annot_proto_from_dir(args.annot_dir)
    save_dir = os.path.dirname(args.save_file)
    if save_dir and not os.path.isdir(save_dir):
        os.makedirs(save_dir)
    with open(args.save_file, 'w') as f:
        json.dump(annot, f, indent=2)
    return 0


if __name__ == '__main__':
    sys.exit(main())
