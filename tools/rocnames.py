"""key / value types of a rocPRIM radix sort kernel from its (demangled) name: the kernels carry
`wrapped_radix_sort_onesweep_config<CONFIG, KEY, VALUE>` - CONFIG is `default_config` or a nested config type (round 4:
csrc/prims.hip tunes the digit width and the tile), so the arguments are split with bracket counting, not a regular expression."""
T = {'unsigned long': 'u64', 'unsigned int': 'u32', 'unsigned short': 'u16', 'unsigned char': 'u8', 'unsigned __int128': 'u128'}


def top_level_args(s, start):
    """arguments of the template whose '<' is at s[start]"""
    depth, cur, out = 0, '', []
    for ch in s[start:]:
        if ch in '<(':
            depth += 1
            if depth == 1:
                continue
        elif ch in '>)':
            depth -= 1
            if depth == 0:
                out.append(cur.strip())
                return out
        if ch == ',' and depth == 1:
            out.append(cur.strip()); cur = ''
        else:
            cur += ch
    return out


def sort_types(name):
    """('u64', 'u32') / ('u64', None) for keys only / None when the name is not an onesweep kernel with known types"""
    tag = 'wrapped_radix_sort_onesweep_config<'
    i = name.find(tag)
    if i < 0:
        return None
    a = top_level_args(name, i + len(tag) - 1)
    if len(a) < 3 or a[-2] not in T:
        return None
    v = a[-1]
    return (T[a[-2]], None if v.endswith('empty_type') else T.get(v))
