"""Association of the TUM RGB-D colour and depth streams by time stamp: the
producer of the `associate.txt` files the input providers read, with the
function names of the reference's slam/preprocess/associate.py:48-131.

The Kinect stamps colour and depth images independently, so every colour stamp
is paired with the closest depth stamp within `max_difference` seconds; a stamp
is used at most once, and closer pairs win over farther ones.

The pairing rule is the reference's (associate.py:70-100): all candidate pairs
with |a - (b + offset)| < max_difference, taken in ascending order of
(difference, a, b), greedily, then sorted by a.  The candidates are found here
by bisection over the sorted second stream instead of the reference's full
cross product, which gives the same candidate set.
"""
import argparse
import bisect
import io
import logging
import os


def read_file_list(filename):
    """'stamp d1 d2 ...' lines -> {float stamp: [d1, d2, ...]} (associate.py:48-68):
    commas and tabs count as blanks, lines starting with '#' and lines with a
    stamp only are skipped, a repeated stamp keeps its last line."""
    out = {}
    with open(filename) as f:
        for line in f.read().replace(",", " ").replace("\t", " ").split("\n"):
            if not line or line[0] == "#":
                continue
            fields = [v.strip() for v in line.split(" ") if v.strip() != ""]
            if len(fields) > 1:
                out[float(fields[0])] = fields[1:]
    return out


def associate(first_list, second_list, offset, max_difference):
    """[(stamp1, stamp2)] sorted by stamp1 (associate.py:70-100)."""
    second = sorted(second_list.keys())
    shifted = [b + offset for b in second]
    candidates = []
    for a in first_list.keys():
        # shifted is ascending: only stamps within the window can qualify; the exact
        # predicate is re-evaluated on the window's neighbours as well
        lo = max(bisect.bisect_left(shifted, a - max_difference) - 1, 0)
        hi = min(bisect.bisect_right(shifted, a + max_difference) + 1, len(second))
        for j in range(lo, hi):
            diff = abs(a - (second[j] + offset))
            if diff < max_difference:
                candidates.append((diff, a, second[j]))
    candidates.sort()
    free_first, free_second = set(first_list.keys()), set(second)
    matches = []
    for _, a, b in candidates:
        if a in free_first and b in free_second:
            free_first.remove(a)
            free_second.remove(b)
            matches.append((a, b))
    matches.sort()
    return matches


def get_association(first_file, second_file):
    """Text of an associate.txt: 'stamp1 data1 stamp2 data2' per match, stamps
    printed with %f (associate.py:102-114)."""
    first_list = read_file_list(first_file)
    second_list = read_file_list(second_file)
    offset = 0.00
    matches = associate(first_list, second_list, offset, max_difference=0.02)
    text = io.StringIO()
    for a, b in matches:
        text.write("%f %s %f %s\n" % (a, " ".join(first_list[a]), b - float(offset), " ".join(second_list[b])))
    return text.getvalue()


def create_association_data(base_dir):
    """Writes <base_dir>/<sequence>/associate.txt from rgb.txt and depth.txt for every
    entry of base_dir (associate.py:120-131)."""
    logger = logging.getLogger("slam_amd")
    for name in os.listdir(base_dir):
        logger.info('Associating data set :%s', name)
        association_file = os.path.join(base_dir, name, 'associate.txt')
        data = get_association(os.path.join(base_dir, name, 'rgb.txt'), os.path.join(base_dir, name, 'depth.txt'))
        with open(association_file, 'w') as fw:
            fw.write(data)
        logger.info('Wrote association data to :%s', association_file)


def main(argv=None):
    """Command line of associate.py:134-162."""
    parser = argparse.ArgumentParser(description="associates two stamped data files")
    parser.add_argument('--first_file', help='first text file (format: timestamp data)')
    parser.add_argument('--second_file', help='second text file (format: timestamp data)')
    parser.add_argument('--first_only', help='only output associated lines from first file', action='store_true')
    parser.add_argument('--offset', help='time offset added to the timestamps of the second file (default: 0.0)',
                        default=0.0)
    parser.add_argument('--max_difference', help='maximally allowed time difference for matching entries '
                                                 '(default: 0.02)', default=0.02)
    parser.add_argument('--base_dir', help='base dir containing the sequence directories')
    args = parser.parse_args(argv)
    if args.base_dir:
        create_association_data(args.base_dir)
        return
    first_list = read_file_list(args.first_file)
    second_list = read_file_list(args.second_file)
    for a, b in associate(first_list, second_list, float(args.offset), float(args.max_difference)):
        if args.first_only:
            print("%f %s" % (a, " ".join(first_list[a])))
        else:
            print("%f %s %f %s" % (a, " ".join(first_list[a]), b - float(args.offset), " ".join(second_list[b])))


if __name__ == '__main__':
    main()
